#!/bin/bash
# Scratch (GPU box): bench frame time of library variants, interleaved: tools/exp_ab_bench.sh "<bench args>" VARIANT...
ARGS=$1; shift
for rep in 1 2 3; do for v in "$@"; do
  if [ "$v" = product ]; then L=""; else L="$PWD/vrenderer_amd/lib/variants/$v/libvrterrain.so"; fi
  echo "$v: $(VRTERRAIN_LIB=$L python3 bench.py --steps 120 --warmup 10 --no-cpu-baseline --no-4k $ARGS 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['kernels']['k_raster']['avg_us'], d['kernels']['k_deferred']['avg_us'])")"
done; done
