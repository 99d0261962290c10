#!/bin/bash
# round 6, call x: branch-free exact-width LayerNorm forward + backward: tests, probe A/B (LAVILA_LN_EXACT=0/1), bench A/B
set -u
O=gpurun_out/r6x
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -q -x 2>&1 | grep -E "^E  |passed|failed|^FAILED" | cut -c1-400 | head > $O/tests.txt
for e in 0 1 0 1; do
  echo "== LAVILA_LN_EXACT=$e" >> $O/rowops.txt
  LAVILA_LN_EXACT=$e timeout 300 python tools/probe_rowops.py 256 20 2>&1 | grep -E "ln_" >> $O/rowops.txt
done
for e in 0 1 0 1; do
  LAVILA_LN_EXACT=$e timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-events 2>/dev/null | grep '^{' | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("LN_EXACT='$e'", d["value"], d["ms_per_step"])' >> $O/ab.txt
done
echo done > $O/finished
