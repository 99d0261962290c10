#!/bin/bash
# round 4, call w: final validation of the tree -- smoke, full GPU suite, default bench (cpu baseline, graphed-step child),
# the self-launched 2-rank rehearsal, rocprofv3 kernel traces of the bench (two streams / one stream)
set -u
O=gpurun_out/r4w
mkdir -p $O
export TMPDIR=/tmp
(timeout 300 python __graft_entry__.py smoke 2>&1 | tail -4) > $O/smoke.log
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -E "^E  |passed|failed|^FAILED|error" | cut -c1-400 | head -40) > $O/pytest.log
(timeout 900 python bench.py 2>$O/bench.err | grep '^{' | tail -1) > $O/bench.json
(timeout 600 python bench.py --gpus 2 --batch 32 --steps 6 --warmup 3 --no-cpu-baseline 2>$O/bench_2rank.err | grep '^{' | tail -1) > $O/bench_2rank_gloo.json
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
echo done > $O/finished
