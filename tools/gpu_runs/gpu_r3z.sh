#!/bin/bash
set -u
O=gpurun_out/r3z
mkdir -p $O
export TMPDIR=/tmp
(timeout 900 python tools/probe_narrator.py --batch 64 --length 77 --half --reps 3 --out $O/narrator_b64.json 2>&1 | tail -3) > $O/probe_b64.log
(timeout 900 python tools/probe_narrator.py --batch 64 --length 77 --returns 10 --sample --half --reps 2 --out $O/narrator_b64_r10.json 2>&1 | tail -3) > $O/probe_r10.log
cd /tmp
for cfg in "n1:--batch 64 --length 30 --half --reps 2 --skip-recompute" "n10:--batch 64 --length 30 --returns 10 --sample --half --reps 2 --skip-recompute"; do
  tag=${cfg%%:*}; args=${cfg#*:}
  (timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_$tag -o nar -- python $GRAFT_REPO_ROOT/tools/probe_narrator.py $args 2>&1 | head -5) > $GRAFT_REPO_ROOT/$O/prof_$tag.log
  DB=$(find $GRAFT_REPO_ROOT/$O/prof_$tag -name "*.db" | head -1)
  [ -n "$DB" ] && python $GRAFT_REPO_ROOT/tools/kernel_stats.py $DB 1 > $GRAFT_REPO_ROOT/$O/kernel_stats_$tag.csv 2>/dev/null
  rm -rf $GRAFT_REPO_ROOT/$O/prof_$tag
done
cd $GRAFT_REPO_ROOT
echo done > $O/finished
