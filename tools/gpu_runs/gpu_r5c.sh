#!/bin/bash
# round 5, call c: rider variants of the time backward kernel (probe), hardware-queue hypothesis for the +3.7 % an RCCL
# communicator costs a single rank (GPU_MAX_HW_QUEUES, creation order of the text side stream), tests touched since call b
set -u
O=gpurun_out/r5c
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python tools/probe_time_bwd_rider.py > $O/time_bwd_rider.txt 2>&1
(timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_graph_step.py tests/test_gpu_stream_attention.py -m gpu -q 2>&1 | grep -E "^E  |passed|failed|^FAILED|rror" | cut -c1-600 | head -40) > $O/pytest.log
B="--steps 10 --warmup 3 --no-cpu-baseline"
export LAVILA_BENCH_GRAPH=0
run() { name=$1; shift; (env "$@" timeout 300 python bench.py $B 2>$O/ab_$name.err | grep '^{' | tail -1) > $O/ab_$name.json; }
run plain X=1
run group_only LAVILA_BENCH_ONE_RANK_RCCL=1 LAVILA_BENCH_NO_DDP=1 LAVILA_DYNAMIC_TILES=0
run group_only_q8 LAVILA_BENCH_ONE_RANK_RCCL=1 LAVILA_BENCH_NO_DDP=1 LAVILA_DYNAMIC_TILES=0 GPU_MAX_HW_QUEUES=8
run group_only_early LAVILA_BENCH_ONE_RANK_RCCL=1 LAVILA_BENCH_NO_DDP=1 LAVILA_DYNAMIC_TILES=0 LAVILA_BENCH_EARLY_TEXT_STREAM=1
run group_ddp LAVILA_BENCH_ONE_RANK_RCCL=1 LAVILA_DYNAMIC_TILES=0
run group_ddp_q8 LAVILA_BENCH_ONE_RANK_RCCL=1 LAVILA_DYNAMIC_TILES=0 GPU_MAX_HW_QUEUES=8
run group_ddp_early LAVILA_BENCH_ONE_RANK_RCCL=1 LAVILA_DYNAMIC_TILES=0 LAVILA_BENCH_EARLY_TEXT_STREAM=1
run group_ddp_dyn_q8 LAVILA_BENCH_ONE_RANK_RCCL=1 GPU_MAX_HW_QUEUES=8
run plain_q8 GPU_MAX_HW_QUEUES=8
run plain_again X=1
python - > $O/ab_summary.txt <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/r5c/ab_*.json')):
    try:
        d = json.load(open(f)); print(os.path.basename(f), d['ms_per_step'], d['value'], d['config'].get('host_enqueue_ms_per_step'), d['config'].get('host_caption_bound'))
    except Exception as e:
        print(os.path.basename(f), 'FAILED', e)
PY
echo done > $O/finished
