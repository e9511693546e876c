#!/bin/bash
# Scratch (GPU box): SQ_INSTS_VALU / SQ_WAVE_CYCLES of k_raster for library variants: tools/pmc_valu.sh VARIANT...
export TMPDIR=/tmp
for v in "$@"; do
  export VARIANT=$v; [ "$v" = "product" ] && unset VARIANT
  rm -rf gpurun_out/pmcv_$v
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d gpurun_out/pmcv_$v --output-format csv -- python3 tools/exp_raster_pmc.py > gpurun_out/pmcv_$v.log 2>&1
  python3 tools/summarize_sq.py gpurun_out/pmcv_$v | grep "k_raster" | sed "s/^/$v /"
done
