#!/bin/bash
# On the GPU box: the bench lines that go with tools/collect_profiles.sh (no profiler), into gpurun_out/prof_$TAG/.
# usage: tools/collect_extras.sh TAG
TAG=$1
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
B="--no-cpu-baseline --no-4k"
line() { grep '^{' $1 | tail -1; }
python3 bench.py --steps 20 --warmup 5 > $OUT/driver_style.log 2>&1; line $OUT/driver_style.log > $OUT/bench_8k_driver_style.json
python3 bench.py $B --steps 120 --warmup 10 > $OUT/unprof.log 2>&1; line $OUT/unprof.log > $OUT/bench_8k_unprofiled.json
python3 bench.py $B --steps 60 --warmup 10 --lights 1024 > $OUT/l.log 2>&1; line $OUT/l.log > $OUT/bench_8k_lights1024_unprofiled.json
python3 bench.py $B --steps 60 --warmup 10 --shadows > $OUT/s.log 2>&1; line $OUT/s.log > $OUT/bench_8k_shadows_unprofiled.json
python3 bench.py $B --steps 120 --warmup 10 --width 3840 --height 2160 > $OUT/4k.log 2>&1; line $OUT/4k.log > $OUT/bench_4k_unprofiled.json
python3 bench.py $B --steps 120 --warmup 10 --fused > $OUT/f.log 2>&1; line $OUT/f.log > $OUT/bench_8k_fused.json
for n in 2 4 8; do
  python3 bench.py $B --steps 120 --warmup 10 --emulate-rank 0 --emulate-world $n > $OUT/e$n.log 2>&1; line $OUT/e$n.log > $OUT/emulated_rank_N$n.json
done
python3 bench.py $B --steps 120 --warmup 10 --emulate-rank 0 --emulate-world 8 --exchange hdr > $OUT/e8h.log 2>&1; line $OUT/e8h.log > $OUT/emulated_rank_N8_hdr.json
rocprofv3 --kernel-trace --stats -d $OUT/stats_rank8 --output-format csv -- python3 bench.py $B --steps 60 --warmup 10 --emulate-rank 0 --emulate-world 8 > $OUT/stats_rank8.log 2>&1
cp $(ls $OUT/stats_rank8/*/*kernel_stats.csv | head -1) $OUT/kernel_stats_emulated_rank_N8.csv; rm -rf $OUT/stats_rank8
python3 - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/*.json")):
    try:
        j = json.loads(open(f).read())
        print(f.split("/")[-1], j.get("value"), j.get("ms_per_step"), (j.get("sustained") or {}).get("value"), j.get("host_issue_ms_per_step"), {k: v["avg_us"] for k, v in j.get("kernels", {}).items() if k in ("k_raster", "k_deferred", "k_raster_lit", "k_deferred_tiled")})
    except Exception as e:
        print(f, "unreadable", e)
PY
