#!/bin/bash
set -u
O=gpurun_out/r3m
mkdir -p $O
export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_gpu_narrator.py -m gpu -q 2>&1 | tail -60 | cut -c1-400) > $O/pytest_narrator.log
(timeout 900 python tools/probe_narrator.py --batch 64 --length 77 --half --reps 3 --out $O/narrator_b64.json 2>&1 | tail -5) > $O/probe_b64.log
cd /tmp
(timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o nar -- python $GRAFT_REPO_ROOT/tools/probe_narrator.py --batch 64 --length 30 --half --reps 2 --skip-recompute 2>&1 | head -40) > $GRAFT_REPO_ROOT/$O/prof.log
cd $GRAFT_REPO_ROOT
DB=$(find $O/prof -name "*.db" | head -1)
[ -n "$DB" ] && python tools/kernel_stats.py $DB 1 > $O/kernel_stats.csv 2>$O/kernel_stats.err
rm -rf $O/prof
echo done > $O/finished
