#!/bin/bash
# Scratch (GPU box): tile-pass time of several library variants on the same device, interleaved twice: tools/ab.sh VARIANT...
for rep in 1 2; do for v in "$@"; do
  if [ "$v" = product ]; then r=$(python3 tools/exp_raster_pmc.py 2>/dev/null | tail -1); else r=$(VARIANT=$v python3 tools/exp_raster_pmc.py 2>/dev/null | tail -1); fi
  echo "$v: $r"
done; done
