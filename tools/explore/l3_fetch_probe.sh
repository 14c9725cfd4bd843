#!/bin/bash
# tools/explore/l3_fetch_probe.sh : FETCH_SIZE / TCC hit+miss of the level-3 kernels at batch 8, 16, 32, 64
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/l3probe
mkdir -p $out
for b in 8 16 32 64; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/f$b -o f -- python3 tools/explore/l3_fetch_probe.py $b > /dev/null 2> $out/f$b.err || exit 1
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $out/t$b -o t -- python3 tools/explore/l3_fetch_probe.py $b > /dev/null 2> $out/t$b.err || echo "tcc pass failed at $b"
  echo "== batch $b"
  python3 tools/pmc_summary.py $out/f$b $out/t$b | grep -A1 "32, 8, 26, 9\|32, 8, 132" 
done
