#!/bin/bash
# Scratch (GPU box): the bench line (N = 1 and the N = 8 rank emulation) for library variants, interleaved: tools/exp_ab_bench.sh VARIANT...
show() { python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
k={n:v['avg_us'] for n,v in j['kernels'].items()}
geo=sum(k.get(n,0) for n in ('k_select','k_vertex','k_setup','k_clip','k_scan','k_fill'))
print('%.3f ms  %.1f Gpx/s  geometry sum %.0f us ' % (j['ms_per_step'], j['value'], geo), k)"; }
for rep in 1 2; do for v in "$@"; do
  if [ "$v" = product ]; then cmd="python3 bench.py"; else cmd="python3 tools/ab_bench.py $v"; fi
  echo -n "$v N=1: "; $cmd --steps 40 --warmup 5 --no-cpu-baseline --no-4k 2>/dev/null | show
  echo -n "$v N=8 rank 0: "; $cmd --emulate-rank 0 --emulate-world 8 --steps 60 --warmup 10 --no-cpu-baseline --no-4k 2>/dev/null | show
done; done
