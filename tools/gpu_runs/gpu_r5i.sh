#!/bin/bash
# round 5, call i: validation of the final tree -- smoke, GPU suite three times (flakiness), default bench (cpu baseline,
# graphed-step child), kernel traces
set -u
O=gpurun_out/r5i
mkdir -p $O
export TMPDIR=/tmp
(timeout 300 python __graft_entry__.py smoke 2>&1 | tail -4) > $O/smoke.log
for i in 1 2 3; do
  (timeout 1800 python -m pytest tests -m gpu -q 2>&1 | grep -E "^E  |passed|failed|^FAILED|rror" | cut -c1-600 | head -30) > $O/pytest_all_$i.log
done
(timeout 900 python bench.py 2>$O/bench.err | grep '^{' | tail -1) > $O/bench.json
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
