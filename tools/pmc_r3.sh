#!/bin/bash
# Scratch (GPU box): SQ + TCP counters of the tile pass alone (8 flythrough frames), product library or VARIANT=...
export TMPDIR=/tmp
run() { n=$1; shift; rm -rf gpurun_out/pmc3_$n; timeout -k 5 180 rocprofv3 --kernel-trace --pmc "$@" -d gpurun_out/pmc3_$n --output-format csv -- python3 tools/exp_raster_pmc.py > gpurun_out/pmc3_$n.log 2>&1 || { echo "pass $n failed"; return 1; }; python3 tools/summarize_sq.py gpurun_out/pmc3_$n | grep -E "k_raster|no counter"; }
run sqa SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR &&
run sqb SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE &&
run tcd TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum &&
run tce TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum &&
run tcc TCP_GATE_EN1_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum &&
run gr GRBM_GUI_ACTIVE
