#!/bin/bash
# usage: tools/run_pmc_ab.sh TAG VARIANT...   (on the GPU box; writes gpurun_out/pmc_TAG_<variant>_{a,b}/)
set -e
TAG=$1; shift
export TMPDIR=/tmp
A="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
B="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
for v in "$@"; do
  export VARIANT=$v
  [ "$v" = "product" ] && unset VARIANT
  rocprofv3 --kernel-trace --pmc $A -d gpurun_out/pmc_${TAG}_${v}_a --output-format csv -- python3 tools/exp_raster_pmc.py > gpurun_out/pmc_${TAG}_${v}_a.log 2>&1
  rocprofv3 --kernel-trace --pmc $B -d gpurun_out/pmc_${TAG}_${v}_b --output-format csv -- python3 tools/exp_raster_pmc.py > gpurun_out/pmc_${TAG}_${v}_b.log 2>&1
done
python3 tools/summarize_sq.py gpurun_out/pmc_${TAG}_*_[ab]
