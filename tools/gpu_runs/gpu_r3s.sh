#!/bin/bash
set -u
O=gpurun_out/r3s
mkdir -p $O
export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_gpu_narrator.py -m gpu -q 2>&1 | tail -60 | cut -c1-500) > $O/pytest_narrator.log
(timeout 900 python tools/probe_narrator.py --batch 64 --length 77 --half --reps 3 --out $O/narrator_b64.json 2>&1 | tail -5) > $O/probe_b64.log
(timeout 900 python tools/probe_narrator.py --batch 64 --length 77 --returns 10 --sample --half --reps 2 --out $O/narrator_b64_r10.json 2>&1 | tail -5) > $O/probe_r10.log
cd /tmp
(timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o nar -- python $GRAFT_REPO_ROOT/tools/probe_narrator.py --batch 64 --length 20 --returns 10 --sample --half --reps 1 --skip-recompute 2>&1 | head -40) > $GRAFT_REPO_ROOT/$O/prof.log
cd $GRAFT_REPO_ROOT
DB=$(find $O/prof -name "*.db" | head -1)
[ -n "$DB" ] && python tools/kernel_stats.py $DB 1 > $O/kernel_stats_r10.csv 2>$O/kernel_stats.err
rm -rf $O/prof
echo done > $O/finished
