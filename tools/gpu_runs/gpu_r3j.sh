#!/bin/bash
set -u
O=gpurun_out/r3j
mkdir -p $O
export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_gpu_narrator.py -m gpu -q 2>&1 | tail -60 | cut -c1-400) > $O/pytest_narrator.log
cd /tmp
(timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o nar -- python $GRAFT_REPO_ROOT/tools/probe_narrator.py --batch 64 --length 30 --half --reps 2 --skip-recompute 2>&1 | tail -30) > $GRAFT_REPO_ROOT/$O/prof.log
cd $GRAFT_REPO_ROOT
f=$(find $O/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -45 "$f" | cut -c1-260 > $O/kernel_stats_head.csv
find $O/prof -name "*.csv" ! -name "*kernel_stats.csv" -delete
find $O/prof -name "*.db" -delete
echo done > $O/finished
