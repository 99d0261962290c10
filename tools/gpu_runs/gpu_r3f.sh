#!/bin/bash
# round-3 GPU pass F: final numbers -- tests, smoke, bench (static / counters), 2-rank rehearsal, kernel trace, PMC traffic
set -u
O=gpurun_out/r3f
mkdir -p $O
export TMPDIR=/tmp
(timeout 1200 python -m pytest tests -m gpu -q 2>&1 | grep -E "^E  |passed|failed|^FAILED" | cut -c1-300 | head -30) > $O/pytest.log
timeout 200 python __graft_entry__.py smoke > $O/smoke.log 2>&1
timeout 300 python bench.py > $O/bench_final.json 2> $O/bench.err
LAVILA_DYNAMIC_TILES=1 timeout 300 python bench.py --no-cpu-baseline > $O/bench_counters.json 2>/dev/null
timeout 400 python bench.py --gpus 2 --batch 32 --steps 4 --warmup 2 > $O/bench_2rank_gloo.json 2> $O/bench_2rank.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-events > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find $O/prof -name "*.db" | head -1)
[ -n "$DB" ] && python tools/kernel_stats.py $DB 5 > $O/kernel_stats.csv 2>$O/kernel_stats.err
rm -rf $O/prof
timeout 900 bash tools/pmc_bench_traffic.sh > $O/traffic.log 2>&1
cp gpurun_out/traffic/r03_*.json $O/ 2>/dev/null
rm -rf gpurun_out/traffic/FETCH_SIZE gpurun_out/traffic/WRITE_SIZE
echo done > $O/finished
