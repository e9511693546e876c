#!/bin/bash
# On the GPU box: rocprofv3 kernel statistics and PMC passes of the bench commands, into gpurun_out/prof_$TAG/.
# usage: tools/collect_profiles.sh TAG [what...]   what = stats pmc sq (default: all)
# afterwards (here): python tools/summarize_pmc.py traffic pmc_fetch_8k.csv pmc_write_8k.csv profiles/rNN_pmc_traffic.json
#                    python tools/summarize_pmc.py sq pmc_sqa_8k.csv pmc_sqb_8k.csv pmc_sqc_8k.csv pmc_sqd_8k.csv profiles/rNN_pmc_sq.json
set -e
TAG=$1; shift
WHAT=${@:-stats pmc sq}
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
B="--no-cpu-baseline --no-4k --no-sustained"     # one timed region: the per-kernel averages are those of the default (tracked) frames only
run_stats() {  # name, bench args
  rocprofv3 --kernel-trace --stats -d $OUT/stats_$1 --output-format csv -- python3 bench.py $B --steps 30 --warmup 3 ${@:2} > $OUT/stats_$1.log 2>&1
  cp $(ls $OUT/stats_$1/*/*kernel_stats.csv | head -1) $OUT/kernel_stats_$1.csv
  grep '^{' $OUT/stats_$1.log | tail -1 > $OUT/bench_$1.json
}
run_pmc() {  # name, counters..., -- bench args
  name=$1; shift; ctrs=(); while [ "$1" != "--" ]; do ctrs+=($1); shift; done; shift
  rocprofv3 --kernel-trace --pmc ${ctrs[@]} -d $OUT/pmc_$name --output-format csv -- python3 bench.py $B --steps 5 --warmup 2 $@ > $OUT/pmc_$name.log 2>&1
  cp $(ls $OUT/pmc_$name/*/*counter_collection.csv | head -1) $OUT/pmc_$name.csv
}
for w in $WHAT; do case $w in
stats)
  run_stats 8k
  run_stats 4k --width 3840 --height 2160
  run_stats 1080p --width 1920 --height 1080
  run_stats 8k_shadows --shadows
  run_stats 8k_lights1024 --lights 1024 ;;
pmc)
  run_pmc fetch_8k FETCH_SIZE --
  run_pmc write_8k WRITE_SIZE --
  run_pmc fetch_4k FETCH_SIZE -- --width 3840 --height 2160
  run_pmc write_4k WRITE_SIZE -- --width 3840 --height 2160
  run_pmc fetch_8k_lights1024 FETCH_SIZE -- --lights 1024
  run_pmc write_8k_lights1024 WRITE_SIZE -- --lights 1024 ;;
sq)
  run_pmc sqa_8k SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --
  run_pmc sqb_8k SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --
  run_pmc sqc_8k SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_SMEM --
  run_pmc sqd_8k GRBM_GUI_ACTIVE --
  run_pmc sqa_8k_lights1024 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -- --lights 1024
  run_pmc sqb_8k_lights1024 SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -- --lights 1024 ;;
esac; done
rm -rf $OUT/stats_*/ $OUT/pmc_*/        # keep the CSV copies only
ls -la $OUT | head -40
