#!/bin/bash
set -u
O=gpurun_out/r3k
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
(timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o nar -- python $GRAFT_REPO_ROOT/tools/probe_narrator.py --batch 64 --length 30 --half --reps 2 --skip-recompute 2>&1 | head -40) > $GRAFT_REPO_ROOT/$O/prof.log
cd $GRAFT_REPO_ROOT
DB=$(find $O/prof -name "*.db" | head -1)
[ -n "$DB" ] && python tools/kernel_stats.py $DB 1 > $O/kernel_stats.csv 2>$O/kernel_stats.err
rm -rf $O/prof
echo done > $O/finished
