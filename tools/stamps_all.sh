#!/bin/bash
# One GPU-box call: stamps of the shipped level-2 / level-3 launches (variants_so/libglomseg_diag.so = a -DGS_DIAG build) and
# their tables.      tools/stamps_all.sh [tag]  -> gpurun_out/stamps_<tag>.txt
tag=${1:-r06}
export GLOMSEG_EXPERIMENT=1 GLOMSEG_ALLOW_DIAG=1 GLOMSEG_LIB=variants_so/libglomseg_diag.so
out=gpurun_out/stamps_$tag.txt
: > $out
# variant  file  NCHUNK CPD  matrix-pipe cycles of one chunk of one wave (k-steps x P x cycles per MFMA)
while read v f nch cpd cyc; do
    GS_VARIANT=$v timeout -k 10 300 python tools/stamps_run.py > gpurun_out/stamps_run_$v.log 2>&1 || { echo "variant $v failed"; tail -5 gpurun_out/stamps_run_$v.log; exit 1; }
    echo "==== GS_VARIANT=$v $f" >> $out
    python tools/stamps3.py gpurun_out/$f $nch $cpd $cyc >> $out 2>&1
done <<'LIST'
160 stamps_l3esp.txt 15 3 4992
164 stamps_l3down.txt 15 3 4992
161 stamps_l2esp.txt 5 1 3456
163 stamps_l2last.txt 5 1 3456
162 stamps_l2down.txt 5 1 3456
LIST
cat $out
