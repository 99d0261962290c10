#!/bin/bash
# round 5, call j: flakiness of the process-group tests after moving the step's collectives to their own stream (8 runs),
# DDP communication hooks on a one-rank group (A/B), full suite once
set -u
O=gpurun_out/r5j
mkdir -p $O
export TMPDIR=/tmp
for i in 1 2 3 4 5 6 7 8; do
  (timeout 600 python -m pytest tests/test_gpu_ddp.py tests/test_gpu_graph_step.py -m gpu -q 2>&1 | grep -E "^E  |passed|failed|^FAILED|rror" | cut -c1-400 | tail -4) > $O/pytest_ddp_$i.log
done
B="--steps 10 --warmup 3 --no-cpu-baseline"
export LAVILA_BENCH_GRAPH=0
run() { name=$1; shift; (env "$@" timeout 300 python bench.py $B 2>$O/ab_$name.err | grep '^{' | tail -1) > $O/ab_$name.json; }
run plain X=1
run ddp_default_scaling LAVILA_BENCH_ONE_RANK_RCCL=1 LAVILA_DYNAMIC_TILES=0 LAVILA_BENCH_DDP_HOOK=none
run ddp_allreduce_hook LAVILA_BENCH_ONE_RANK_RCCL=1 LAVILA_DYNAMIC_TILES=0 LAVILA_BENCH_DDP_HOOK=allreduce
run ddp_bf16_hook LAVILA_BENCH_ONE_RANK_RCCL=1 LAVILA_DYNAMIC_TILES=0 LAVILA_BENCH_DDP_HOOK=bf16
run ddp_default_scaling2 LAVILA_BENCH_ONE_RANK_RCCL=1 LAVILA_DYNAMIC_TILES=0 LAVILA_BENCH_DDP_HOOK=none
run ddp_allreduce_hook2 LAVILA_BENCH_ONE_RANK_RCCL=1 LAVILA_DYNAMIC_TILES=0 LAVILA_BENCH_DDP_HOOK=allreduce
run plain2 X=1
python - > $O/ab_summary.txt <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/r5j/ab_*.json')):
    try:
        d = json.load(open(f)); print(os.path.basename(f), d['ms_per_step'], d['value'], d['roofline']['frac'], d['config'].get('host_enqueue_ms_per_step'))
    except Exception as e:
        print(os.path.basename(f), 'FAILED', e)
PY
(timeout 1800 python -m pytest tests -m gpu -q 2>&1 | grep -E "^E  |passed|failed|^FAILED|rror" | cut -c1-600 | head -30) > $O/pytest_all.log
echo done > $O/finished
