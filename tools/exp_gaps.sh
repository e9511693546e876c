#!/bin/bash
# Scratch (GPU box): idle time between the tile pass and the lighting pass for bench variants (timing level, per-call API, no prepare)
export TMPDIR=/tmp
B="--no-cpu-baseline --no-4k --no-sustained --steps 60 --warmup 10"
run() { n=$1; shift; rm -rf gpurun_out/gaps_$n; rocprofv3 --kernel-trace -d gpurun_out/gaps_$n --output-format csv -- python3 bench.py $B "$@" > gpurun_out/gaps_$n.log 2>&1; echo "== $n: $@"; grep '^{' gpurun_out/gaps_$n.log | python3 -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'], j['host_issue_ms_per_step'])"; python3 tools/trace_gaps.py gpurun_out/gaps_$n ${WINDOW:+--window}; rm -rf gpurun_out/gaps_$n; }
run t2
run t0 --timing-level 0
run t2_nosubmit --no-submit
run t0_nosubmit --timing-level 0 --no-submit
run t0_noprep --timing-level 0 --no-prepare
run n8 --emulate-rank 0 --emulate-world 8 --exchange hdr
run n8_t0 --emulate-rank 0 --emulate-world 8 --exchange hdr --timing-level 0
