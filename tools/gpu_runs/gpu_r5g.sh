#!/bin/bash
# round 5, call g: embedding backward kernel (test + A/B in the bench), GPU_MAX_HW_QUEUES set from inside bench.py
set -u
O=gpurun_out/r5g
mkdir -p $O
export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -q 2>&1 | grep -E "^E  |passed|failed|^FAILED|rror" | cut -c1-600 | head -40) > $O/pytest.log
B="--steps 10 --warmup 3 --no-cpu-baseline"
export LAVILA_BENCH_GRAPH=0
run() { name=$1; shift; (env "$@" timeout 300 python bench.py $B 2>$O/ab_$name.err | grep '^{' | tail -1) > $O/ab_$name.json; }
run plain_a X=1
run group_only_inproc_q8 LAVILA_BENCH_ONE_RANK_RCCL=1 LAVILA_BENCH_NO_DDP=1 LAVILA_DYNAMIC_TILES=0
run group_only_q4 LAVILA_BENCH_ONE_RANK_RCCL=1 LAVILA_BENCH_NO_DDP=1 LAVILA_DYNAMIC_TILES=0 GPU_MAX_HW_QUEUES=4
run plain_b X=1
python - > $O/ab_summary.txt <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/r5g/ab_*.json')):
    try:
        d = json.load(open(f)); print(os.path.basename(f), d['ms_per_step'], d['value'], d['roofline']['frac'], d['config'].get('host_enqueue_ms_per_step'))
    except Exception as e:
        print(os.path.basename(f), 'FAILED', e)
PY
cd /tmp
LAVILA_TEXT_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_serial -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-events > $GRAFT_REPO_ROOT/$O/prof_serial.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find $O/prof_serial -name "*.db" | head -1)
[ -n "$DB" ] && python tools/kernel_stats.py $DB 7 > $O/bench_kernel_stats_serial.csv 2>$O/kernel_stats_serial.err
rm -rf $O/prof_serial
echo done > $O/finished
