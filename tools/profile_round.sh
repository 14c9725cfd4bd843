#!/bin/bash
# One GPU-box call that regenerates everything under profiles/ for a round:
#   tools/profile_round.sh <tag>      (run through gpurun from the repo root; writes gpurun_out/prof_<tag>/)
# rocprofv3 runs the interpreter directly after `--` (no env / bash -c hop) and the counter passes are
# separate from the kernel trace, as the pool requires.
set -o pipefail
tag=${1:-rXX}
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
# one batch in flight under the profiler: with two, co-running kernels stretch each other's durations in the trace
B="python3 bench.py --steps 20 --warmup 3 --repeats 1 --lanes 1 --no-cpu-baseline --no-host-pipeline --no-real-crops"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o stats -- $B > $out/bench_under_rocprof.json 2> $out/stats.err || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -o fetch -- $B > /dev/null 2> $out/fetch.err || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -o write -- $B > /dev/null 2> $out/write.err || exit 1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/sq -o sq -- $B > /dev/null 2> $out/sq.err || exit 1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS --output-format csv -d $out/sq2 -o sq2 -- $B > /dev/null 2> $out/sq2.err || true
python3 tools/pmc_summary.py $out/fetch $out/write $out/sq $out/sq2 > $out/pmc_summary.txt
find $out/stats -name '*kernel_stats.csv' -exec cp {} $out/kernel_stats.csv \;
python3 tools/make_traffic.py $out/pmc_summary.txt $tag > $out/traffic.json
python3 bench.py > $out/bench.json 2> $out/bench.err
cat $out/bench.json
