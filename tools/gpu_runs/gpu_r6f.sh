#!/bin/bash
# round 6, call f: the whole GPU suite + smoke on the round's tree, the default bench line, kernel traces (two streams / serial),
# PMC traffic passes of the bench step, PMC of the fused space backward, one-rank RCCL record, 2-rank rehearsal
set -u
O=gpurun_out/r6f
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python tools/probe_space_bwd_ab.py 2>&1 | grep -v amdgpu.ids > $O/space_bwd_ab.txt
timeout 300 python tools/probe_space_bwd_ab.py time 2>&1 | grep -v amdgpu.ids >> $O/space_bwd_ab.txt
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | cut -c1-300 > $O/tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -6 | cut -c1-400 > $O/smoke.txt
(timeout 400 python bench.py 2>$O/bench.err | grep '^{' | tail -1) > $O/bench.json
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-events > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find $O/prof -name "*.db" | head -1)
[ -n "$DB" ] && python tools/kernel_stats.py $DB 7 > $O/bench_kernel_stats_final.csv 2>$O/kernel_stats.err
rm -rf $O/prof
cd /tmp
LAVILA_TEXT_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-events > $GRAFT_REPO_ROOT/$O/prof_serial.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find $O/prof -name "*.db" | head -1)
[ -n "$DB" ] && python tools/kernel_stats.py $DB 7 > $O/bench_kernel_stats_serial.csv 2>>$O/kernel_stats.err
rm -rf $O/prof
timeout 900 bash tools/pmc_bench_traffic.sh > $O/traffic.log 2>&1
cp gpurun_out/traffic/r06_traffic_*.json $O/ 2>/dev/null
timeout 600 bash tools/pmc_probe.sh space_bwd_r6 space bwd 256 3 > $O/pmc_space_bwd.log 2>&1
cp gpurun_out/pmc_space_bwd_r6/summary.txt $O/pmc_space_bwd_fused.txt 2>/dev/null
rm -rf gpurun_out/traffic/FETCH_SIZE gpurun_out/traffic/WRITE_SIZE gpurun_out/pmc_space_bwd_r6/p*
(LAVILA_BENCH_ONE_RANK_RCCL=1 timeout 400 python bench.py --no-cpu-baseline 2>$O/one_rank.err | grep '^{' | tail -1) > $O/bench_one_rank_rccl.json
(timeout 600 python bench.py --gpus 2 --batch 32 --steps 4 --warmup 1 --no-cpu-baseline 2>$O/two_rank.err | grep '^{' | tail -1) > $O/bench_2rank_gloo.json
for f in $O/*.err; do tail -c 1200 $f > $f.tail; rm $f; done
echo done > $O/finished
