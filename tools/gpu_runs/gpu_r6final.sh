#!/bin/bash
# round 6, final records of the second session: GPU suite, smoke, default bench line, kernel traces (two streams / serial), PMC
# traffic passes, one-rank RCCL, config 4, narrator, configs[2] checkpointed, 2-rank gloo rehearsal
set -u
O=gpurun_out/r6final
mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | grep -E "^E  |passed|failed|^FAILED" | cut -c1-300 | head -20 > $O/tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -6 | cut -c1-400 > $O/smoke.txt
(timeout 600 python bench.py 2>$O/bench.err | grep '^{' | tail -1) > $O/bench.json
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-events > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find $O/prof -name "*.db" | head -1)
[ -n "$DB" ] && python tools/kernel_stats.py $DB 7 > $O/bench_kernel_stats.csv 2>$O/kernel_stats.err
rm -rf $O/prof
cd /tmp
LAVILA_TEXT_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-events > $GRAFT_REPO_ROOT/$O/prof_serial.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find $O/prof -name "*.db" | head -1)
[ -n "$DB" ] && python tools/kernel_stats.py $DB 7 > $O/bench_kernel_stats_serial.csv 2>>$O/kernel_stats.err
rm -rf $O/prof
timeout 900 bash tools/pmc_bench_traffic.sh > $O/traffic.log 2>&1
cp gpurun_out/traffic/r06_traffic_*.json $O/ 2>/dev/null
rm -rf gpurun_out/traffic/FETCH_SIZE gpurun_out/traffic/WRITE_SIZE
(LAVILA_BENCH_ONE_RANK_RCCL=1 timeout 400 python bench.py --no-cpu-baseline 2>$O/one_rank.err | grep '^{' | tail -1) > $O/bench_one_rank_rccl.json
(timeout 600 python bench.py --model CLIP_OPENAI_TIMESFORMER_LARGE_336PX --frames 16 --batch 24 --steps 4 --warmup 2 --no-cpu-baseline 2>$O/config4.err | grep '^{' | tail -1) > $O/bench_config4.json
(timeout 600 python bench.py --model CLIP_OPENAI_TIMESFORMER_LARGE_336PX --frames 16 --batch 8 --steps 4 --warmup 2 --no-cpu-baseline 2>>$O/config4.err | grep '^{' | tail -1) > $O/bench_config4_b8.json
(timeout 600 python bench.py --workload narrator --no-cpu-baseline 2>$O/narrator.err | grep '^{' | tail -1) > $O/bench_narrator.json
(timeout 900 python bench.py --frames 16 --batch 256 --checkpoint --steps 3 --warmup 1 --no-cpu-baseline 2>$O/config3.err | grep '^{' | tail -1) > $O/bench_config3_b256_16f_ckpt.json
(timeout 900 python bench.py --gpus 2 --steps 4 --warmup 2 --no-cpu-baseline 2>$O/gloo2.err | grep '^{' | tail -1) > $O/bench_2rank_gloo.json
echo done > $O/finished
