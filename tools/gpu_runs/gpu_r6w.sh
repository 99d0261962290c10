#!/bin/bash
# round 6, call w: branch-free exact-width LayerNorm forward (no conservative vmcnt(0) in the row loop): tests + probe A/B
set -u
O=gpurun_out/r6w
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "layernorm or add_layer" 2>&1 | grep -E "^E  |passed|failed|^FAILED" | cut -c1-400 | head > $O/tests.txt
for e in 0 1 0 1; do
  echo "== LAVILA_LN_EXACT=$e" >> $O/rowops.txt
  LAVILA_LN_EXACT=$e timeout 300 python tools/probe_rowops.py 256 20 2>&1 | grep -E "ln_fwd" >> $O/rowops.txt
done
echo done > $O/finished
