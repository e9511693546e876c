#!/bin/bash
# Scratch (GPU box): texture-address / L1 counters of the tile pass, two counters of a block per pass (four TA counters
# in one pass abort rocprofv3 and leave the run hanging): tools/pmc_ta.sh [VARIANT]
export TMPDIR=/tmp
[ -n "$1" ] && export VARIANT=$1
run() { n=$1; shift; rm -rf gpurun_out/pmct_$n; timeout -k 5 120 rocprofv3 --kernel-trace --pmc "$@" -d gpurun_out/pmct_$n --output-format csv -- python3 tools/exp_raster_pmc.py > gpurun_out/pmct_$n.log 2>&1 || { echo "pass $n failed"; return 1; }; python3 tools/summarize_sq.py gpurun_out/pmct_$n | grep -E "k_raster|no counter"; }
run b TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum &&
run c TCP_GATE_EN1_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum &&
run d TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum &&
run e TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum &&
run f TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum &&
run g TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
