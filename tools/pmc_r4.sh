#!/bin/bash
# Scratch (GPU box): per-class vector-instruction counters of the tile pass alone (8 flythrough frames), product library or VARIANT=...
# Round 4: the dynamic class mix that the calibrated issue costs (profiles/r04_valu_issue_costs.txt) are applied to.
export TMPDIR=/tmp
run() { n=$1; shift; rm -rf gpurun_out/pmc4_$n; timeout -k 5 180 rocprofv3 --kernel-trace --pmc "$@" -d gpurun_out/pmc4_$n --output-format csv -- python3 tools/exp_raster_pmc.py > gpurun_out/pmc4_$n.log 2>&1 || { echo "pass $n failed"; return 1; }; python3 tools/summarize_sq.py gpurun_out/pmc4_$n | grep -E "k_raster|no counter"; }
run cls SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT &&
run ins SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES &&
run act SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM_RD &&
run gr GRBM_GUI_ACTIVE
