#!/bin/bash
# round 6, call ab: every stale weight copy re-cast in one launch at the top of the training forward (lvl_cast_transpose_multi):
# tests (kernels, model, DDP / ZeRO, graph step), bench A/B LAVILA_WEIGHT_REFRESH=0 / 1
set -u
O=gpurun_out/r6ab
mkdir -p $O
export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_ddp.py tests/test_gpu_graph_step.py tests/test_gpu_boundary.py -q 2>&1 | grep -E "^E  |passed|failed|^FAILED" | cut -c1-400 | head > $O/tests.txt
for e in 0 1 0 1; do
  LAVILA_WEIGHT_REFRESH=$e timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-events 2>/dev/null | grep '^{' | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("REFRESH='$e'", d["value"], d["ms_per_step"], d["config"]["final_loss"])' >> $O/ab.txt
done
echo done > $O/finished
