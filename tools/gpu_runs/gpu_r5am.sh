#!/bin/bash
# round 5, call am: default bench with the host-side caption bound in the timed loop + kernel trace of the final tree
set -u
O=gpurun_out/r5am
mkdir -p $O
export TMPDIR=/tmp
timeout 240 python bench.py > $O/bench.json 2> $O/bench.err
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_am -o am -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-events > $GRAFT_REPO_ROOT/$O/prof_bench.json 2> $GRAFT_REPO_ROOT/$O/prof.err
cd $GRAFT_REPO_ROOT
f=$(find /tmp/prof_am -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" $O/kernel_stats.csv
echo done > $O/finished
