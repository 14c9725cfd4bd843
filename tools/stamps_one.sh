#!/bin/bash
# tools/stamps_one.sh <lib> <variant> <file> <nchunk> <cpd> <cycles> : one stamped run + table
export GLOMSEG_EXPERIMENT=1 GLOMSEG_ALLOW_DIAG=1 GLOMSEG_LIB=$1
GS_VARIANT=$2 timeout -k 10 300 python tools/stamps_run.py > gpurun_out/stamps_run_$2.log 2>&1 || { echo "variant $2 failed"; tail -5 gpurun_out/stamps_run_$2.log; exit 1; }
echo "==== $1 GS_VARIANT=$2"
python tools/stamps3.py gpurun_out/$3 $4 $5 $6
