#!/bin/bash
# Usage (on the GPU box, via gpurun): tools/pmc_mem_passes.sh <tag> <python script> [args...]
# Memory-side counters (L2 = TCC, vector L1 = TCP, address / data units = TA / TD), one rocprofv3 run per pass so that
# no pass asks for more than the hardware holds (4 TCC, 4 TCP, 2 TA, 2 TD).  Output under gpurun_out/<tag>/.
set -e
tag=$1; shift
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
out=gpurun_out/$tag
mkdir -p $out
run() { name=$1; shift; timeout -s KILL 120 rocprofv3 "$@" --output-format csv -d $out/$name -- python3 $ARGS > $out/$name.log 2>&1; }
ARGS="$*"
run trace --kernel-trace --stats
run tcc_a --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum
run tcc_b --pmc TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum TCC_EA_RDREQ_DRAM_sum TCC_EA_RDREQ_LEVEL_sum
run tcc_c --pmc TCC_EA_WRREQ_sum TCC_EA_WRREQ_64B_sum TCC_EA_WRREQ_STALL_sum TCC_EA_WRREQ_LEVEL_sum
run tcc_d --pmc TCC_BUSY_sum TCC_CYCLE_sum TCC_READ_sum TCC_WRITE_sum
run tcc_e --pmc TCC_STREAMING_REQ_sum TCC_NC_REQ_sum TCC_UC_REQ_sum TCC_CC_REQ_sum
run tcc_f --pmc TCC_IB_STALL_sum TCC_SRC_FIFO_FULL_sum TCC_LATENCY_FIFO_FULL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum
run tcc_g --pmc TCC_EA_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA_WRREQ_DRAM_CREDIT_STALL_sum TCC_NORMAL_WRITEBACK_sum TCC_NORMAL_EVICT_sum
run tcp_a --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
run tcp_b --pmc TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN1_sum
run tcp_c --pmc TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_TOTAL_READ_sum TCP_TOTAL_WRITE_sum
run tcp_d --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_LFIFO_STALL_CYCLES_sum
run ta_a --pmc TA_BUSY_avr TA_TA_BUSY_sum TD_TD_BUSY_sum TD_TC_STALL_sum
run ta_b --pmc TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TD_LOAD_WAVEFRONT_sum TD_STORE_WAVEFRONT_sum
run ta_c --pmc TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum TD_COALESCABLE_WAVEFRONT_sum TD_SPI_STALL_sum
run grbm --pmc GRBM_GUI_ACTIVE
echo passes done
