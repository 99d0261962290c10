#!/bin/bash
# round-3 GPU pass D: same-box A/B matrix + a kernel trace without the text side stream (uninflated durations)
set -u
O=gpurun_out/r3d
mkdir -p $O
export TMPDIR=/tmp
run() { # tag, env...
  tag=$1; shift
  env "$@" timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_$tag.json
  python -c "import json,sys; d=json.load(open('$O/bench_$tag.json')); print('$tag', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline_wgrad']['achieved'])" >> $O/ab.txt
}
run default A=1
run dyn0 LAVILA_DYNAMIC_TILES=0
run smallgrid LAVILA_SMALL_GRID=1
run smallgrid_dyn0 LAVILA_SMALL_GRID=1 LAVILA_DYNAMIC_TILES=0
run default2 A=1
cd /tmp
LAVILA_TEXT_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-events > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find $O/prof -name "*.db" | head -1)
[ -n "$DB" ] && python tools/kernel_stats.py $DB 5 > $O/kernel_stats_serial.csv 2>$O/kernel_stats.err
rm -rf $O/prof
echo done > $O/finished
