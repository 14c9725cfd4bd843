#!/bin/bash
# Vector-memory path counters (TA / TCP / TD / SQ fifo) per kernel, a few per pass: which unit between the
# SIMDs and the L2 is the busy one.  usage (through gpurun): tools/pmc_mem_path.sh <tag>
tag=${1:-mem}
out=gpurun_out/pmc_$tag
mkdir -p $out
export TMPDIR=/tmp
B="python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline"
i=0
while read -r line; do
    [ -z "$line" ] && continue
    i=$((i + 1))
    rocprofv3 --pmc $line --output-format csv -d $out/p$i -o p$i -- $B > /dev/null 2> $out/p$i.err || echo "pass $i ($line) failed" >> $out/failed.txt
done <<'LIST'
TA_TA_BUSY_sum TA_BUFFER_TOTAL_CYCLES_sum GRBM_GUI_ACTIVE
TA_BUFFER_READ_WAVEFRONTS_sum TA_BUFFER_WRITE_WAVEFRONTS_sum
TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_BUFFER_COALESCED_READ_CYCLES_sum
TCP_GATE_EN1_sum TCP_GATE_EN2_sum
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
TCP_TA_TCP_STATE_READ_sum TCP_TCC_READ_REQ_LATENCY_sum
TD_TD_BUSY_sum TD_TC_STALL_sum
TD_LOAD_WAVEFRONT_sum TD_SPI_STALL_sum
SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_WAVE_CYCLES
SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_BUSY_CU_CYCLES
LIST
python3 tools/pmc_summary.py $out/p* > $out/summary.txt
cat $out/failed.txt 2>/dev/null
wc -l $out/summary.txt
