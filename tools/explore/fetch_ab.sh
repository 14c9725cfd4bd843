#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of every kernel for the shipped library and one variant (variants_so/libglomseg_<name>.so):
#   tools/explore/fetch_ab.sh <name>     -> gpurun_out/fetch_ab_<name>.txt
# (the library is chosen by environment variables exported BEFORE rocprofv3 starts the interpreter)
name=$1
out=gpurun_out/fetch_ab_$name
mkdir -p $out
export TMPDIR=/tmp
B="python3 bench.py --steps 10 --warmup 2 --repeats 1 --lanes 1 --no-cpu-baseline --no-host-pipeline --no-real-crops"
for which in shipped $name; do
  if [ $which != shipped ]; then export GLOMSEG_EXPERIMENT=1 GLOMSEG_ALLOW_DIAG=1 GLOMSEG_LIB=variants_so/libglomseg_$name.so; fi
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/${which}_f -o f -- $B > /dev/null 2> $out/${which}_f.err || exit 1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/${which}_w -o w -- $B > /dev/null 2> $out/${which}_w.err || exit 1
  python3 tools/pmc_summary.py $out/${which}_f $out/${which}_w > $out/$which.txt
done
for which in shipped $name; do echo "== $which"; grep -A1 "conv_mfma_kernel<32, 8, 26, 9, 1, 5" $out/$which.txt; done > gpurun_out/fetch_ab_$name.txt
cat gpurun_out/fetch_ab_$name.txt
