#!/bin/bash
# round 5, call f: measurement set of the round's tree -- smoke, GPU suite, default bench (cpu baseline, graphed-step child),
# kernel traces (two streams / one stream), PMC traffic passes of the bench step, PMC of the fused space backward, narrator
# bench lines, config-4 and 16-frame bench lines, 2-rank rehearsal
set -u
O=gpurun_out/r5f
mkdir -p $O
export TMPDIR=/tmp
(timeout 300 python __graft_entry__.py smoke 2>&1 | tail -4) > $O/smoke.log
(timeout 1800 python -m pytest tests -m gpu -q 2>&1 | grep -E "^E  |passed|failed|^FAILED|rror" | cut -c1-600 | head -40) > $O/pytest_all.log
(timeout 900 python bench.py 2>$O/bench.err | grep '^{' | tail -1) > $O/bench.json
(timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>$O/bench20.err | grep '^{' | tail -1) > $O/bench_20steps.json
(timeout 600 python bench.py --gpus 2 --batch 32 --steps 6 --warmup 3 --no-cpu-baseline 2>$O/bench_2rank.err | grep '^{' | tail -1) > $O/bench_2rank_gloo.json
(LAVILA_BENCH_ONE_RANK_RCCL=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>$O/bench_one_rank.err | grep '^{' | tail -1) > $O/bench_one_rank_rccl.json
(timeout 600 python bench.py --model CLIP_OPENAI_TIMESFORMER_LARGE_336PX --frames 16 --batch 8 --steps 6 --warmup 2 --no-cpu-baseline 2>$O/bench_config4.err | grep '^{' | tail -1) > $O/bench_config4.json
(timeout 600 python bench.py --frames 16 --batch 64 --steps 6 --warmup 2 --no-cpu-baseline 2>$O/bench_16f.err | grep '^{' | tail -1) > $O/bench_config3_shape_16f.json
(timeout 600 python bench.py --workload narrator --returns 1 --steps 4 --warmup 1 2>$O/nar1.err | grep '^{' | tail -1) > $O/bench_narrator_n1.json
(timeout 600 python bench.py --workload narrator --returns 10 --steps 4 --warmup 1 2>$O/nar10.err | grep '^{' | tail -1) > $O/bench_narrator_n10.json
for mode in default serial; do
  cd /tmp
  if [ $mode = serial ]; then export LAVILA_TEXT_STREAM=0; else unset LAVILA_TEXT_STREAM; fi
  timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_$mode -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-events > $GRAFT_REPO_ROOT/$O/prof_$mode.log 2>&1
  cd $GRAFT_REPO_ROOT
  DB=$(find $O/prof_$mode -name "*.db" | head -1)
  [ -n "$DB" ] && python tools/kernel_stats.py $DB 7 > $O/bench_kernel_stats_$mode.csv 2>$O/kernel_stats_$mode.err
  rm -rf $O/prof_$mode
done
unset LAVILA_TEXT_STREAM
timeout 900 bash tools/pmc_bench_traffic.sh > $O/traffic.log 2>&1
cp gpurun_out/traffic/r05_traffic_*.json $O/ 2>/dev/null
timeout 600 bash tools/pmc_probe.sh space_bwd_r5 space bwd 256 3 > $O/pmc_space_bwd.log 2>&1
cp gpurun_out/pmc_space_bwd_r5/summary.txt $O/pmc_space_bwd_fused.txt 2>/dev/null
rm -rf gpurun_out/traffic/FETCH_SIZE gpurun_out/traffic/WRITE_SIZE gpurun_out/pmc_space_bwd_r5/p*
echo done > $O/finished
