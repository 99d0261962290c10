#!/bin/bash
# round 5, call b: re-tuned spread fixtures (1.5x / 2x), bias-gradient riders (dq column sums inside the fused space / time
# backward kernels, column-sum tokens for the v third), A/B of the bench step with the tokens on / off, second bisect of the
# process-group overhead (gloo group, lazy RCCL init, monitoring off)
set -u
O=gpurun_out/r5b
mkdir -p $O
export TMPDIR=/tmp
(timeout 1800 python -m pytest tests -m gpu -q -s 2>&1 | grep -E "^E  |passed|failed|^FAILED|rror|f32-class|fp8 end" | cut -c1-900 | head -100) > $O/pytest_all.log
B="--steps 10 --warmup 3 --no-cpu-baseline"
export LAVILA_BENCH_GRAPH=0
run() { name=$1; shift; (env "$@" timeout 300 python bench.py $B 2>$O/ab_$name.err | grep '^{' | tail -1) > $O/ab_$name.json; }
run tokens_on X=1
run tokens_off LAVILA_COLSUM_TOKENS=0
run tokens_on2 X=1
run tokens_off2 LAVILA_COLSUM_TOKENS=0
run group_only_static LAVILA_BENCH_ONE_RANK_RCCL=1 LAVILA_BENCH_NO_DDP=1 LAVILA_DYNAMIC_TILES=0
run group_gloo LAVILA_BENCH_ONE_RANK_RCCL=1 LAVILA_BENCH_NO_DDP=1 LAVILA_DYNAMIC_TILES=0 LAVILA_BENCH_GROUP_BACKEND=gloo
run group_lazy LAVILA_BENCH_ONE_RANK_RCCL=1 LAVILA_BENCH_NO_DDP=1 LAVILA_DYNAMIC_TILES=0 LAVILA_BENCH_LAZY_INIT=1
run group_nomon LAVILA_BENCH_ONE_RANK_RCCL=1 LAVILA_BENCH_NO_DDP=1 LAVILA_DYNAMIC_TILES=0 TORCH_NCCL_ENABLE_MONITORING=0 TORCH_NCCL_ASYNC_ERROR_HANDLING=0
run group_ddp_static LAVILA_BENCH_ONE_RANK_RCCL=1 LAVILA_DYNAMIC_TILES=0
run group_ddp_static_nobucketview LAVILA_BENCH_ONE_RANK_RCCL=1 LAVILA_DYNAMIC_TILES=0 LAVILA_BENCH_NO_BUCKET_VIEW=1
python - > $O/ab_summary.txt <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/r5b/ab_*.json')):
    try:
        d = json.load(open(f)); print(os.path.basename(f), d['ms_per_step'], d['value'], d['config'].get('host_enqueue_ms_per_step'))
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
