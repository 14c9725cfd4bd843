#!/bin/bash
# tools/explore/l3_fetch_variant.sh name ... : FETCH_SIZE of the level-3 kernels at batch 32 for variants_so/libglomseg_<name>.so
# (the library is chosen by the GLOMSEG_LIB environment variable, exported before rocprofv3 starts the interpreter)
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/l3probe
mkdir -p $out
for v in "$@"; do
  export GLOMSEG_EXPERIMENT=1 GLOMSEG_ALLOW_DIAG=1 GLOMSEG_LIB=variants_so/libglomseg_$v.so
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/v_$v -o f -- python3 tools/explore/l3_fetch_probe.py 32 > /dev/null 2> $out/v_$v.err || exit 1
  echo "== $v"
  python3 tools/pmc_summary.py $out/v_$v | grep -A1 "32, 8, 26, 9"
done
