#!/bin/bash
# round 4, call af: kernel trace of the config-4 bench step (TSF-L/14@336 x 16 frames, batch 8) and PMC passes of the fused
# space backward at the TSF-B shape on the final tree
set -u
O=gpurun_out/r4af
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_c4 -o b -- python $GRAFT_REPO_ROOT/bench.py --model CLIP_OPENAI_TIMESFORMER_LARGE_336PX --frames 16 --batch 8 --steps 4 --warmup 2 --no-cpu-baseline --no-events > $GRAFT_REPO_ROOT/$O/prof_c4.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find $O/prof_c4 -name "*.db" | head -1)
[ -n "$DB" ] && python tools/kernel_stats.py $DB 6 > $O/config4_kernel_stats.csv 2>$O/kernel_stats.err
rm -rf $O/prof_c4
timeout 400 bash tools/pmc_probe.sh space_bwd_r4af space bwd 256 3 > $O/pmc_bwd.log 2>&1
cp gpurun_out/pmc_space_bwd_r4af/summary.txt $O/pmc_space_bwd_summary.txt 2>/dev/null
echo done > $O/finished
