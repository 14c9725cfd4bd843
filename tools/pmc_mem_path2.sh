#!/bin/bash
# Vector-memory path counters of the current build, one lane (round 6): TA / TD / TCP busy and stall cycles per kernel.
# (The TA_BUFFER_* counters of tools/pmc_mem_path.sh are NOT in the list: on this stack that pass aborts rocprofv3 with signal 6 and the
# run sits until it is killed -- round 6 lost seven GPU-minutes to it.  Only pass 1 of this list was collected in round 6: TA busy 73 %,
# TD busy 89 % of the dominant launch -- "busy" includes waiting for the L1 / L2, so a latency figure, not a throughput one.)
#   tools/pmc_mem_path2.sh <tag> [lib]      (through gpurun; rocprofv3 runs python3 directly)
tag=${1:-mem2}
out=gpurun_out/pmc_$tag
mkdir -p $out
export TMPDIR=/tmp
if [ -n "$2" ]; then export GLOMSEG_EXPERIMENT=1 GLOMSEG_ALLOW_DIAG=1 GLOMSEG_LIB=$2; fi
B="python3 bench.py --steps 6 --warmup 2 --repeats 1 --lanes 1 --no-cpu-baseline --no-host-pipeline --no-real-crops"
i=0
while read -r line; do
    [ -z "$line" ] && continue
    i=$((i + 1))
    rocprofv3 --pmc $line --output-format csv -d $out/p$i -o p$i -- $B > /dev/null 2> $out/p$i.err || echo "pass $i ($line) failed" >> $out/failed.txt
done <<'LIST'
TA_TA_BUSY_sum TD_TD_BUSY_sum GRBM_GUI_ACTIVE
TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TD_TC_STALL_sum
SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES
SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum
LIST
python3 tools/pmc_summary.py $out/p* > $out/summary.txt
cat $out/failed.txt 2>/dev/null
grep -A1 "conv_mfma_kernel<32, 8, 26, 9, 1, 5, 28, 25, 2, 13, 4989955>\|4, 9, 287747>\|4, 9, 279553>\|4, 9, 41999>\|dec_tail\|2, 1, 12, 12, 8, 3" $out/summary.txt
