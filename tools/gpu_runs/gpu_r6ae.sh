#!/bin/bash
# round 6, call ae: kernel trace of config 4 (TSF-L/14 @336, 16 frames, local batch 8) on the last tree
set -u
O=gpurun_out/r6ae
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
LAVILA_TEXT_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o b -- python $GRAFT_REPO_ROOT/bench.py --model CLIP_OPENAI_TIMESFORMER_LARGE_336PX --frames 16 --batch 8 --steps 3 --warmup 1 --no-cpu-baseline --no-events > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find $O/prof -name "*.db" | head -1)
[ -n "$DB" ] && python tools/kernel_stats.py $DB 4 > $O/config4_kernel_stats_serial.csv 2>$O/kernel_stats.err
rm -rf $O/prof
echo done > $O/finished
