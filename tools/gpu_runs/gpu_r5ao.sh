#!/bin/bash
# round 5, call ao: the default bench line and the kernel trace of the last tree
set -u
O=gpurun_out/r5ao
mkdir -p $O
export TMPDIR=/tmp
(timeout 300 python bench.py 2>$O/bench.err | grep '^{' | tail -1) > $O/bench.json
cd /tmp
timeout 240 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-events > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find $O/prof -name "*.db" | head -1)
[ -n "$DB" ] && python tools/kernel_stats.py $DB 7 > $O/bench_kernel_stats.csv 2>$O/kernel_stats.err
rm -rf $O/prof
echo done > $O/finished
