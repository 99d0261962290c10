#!/bin/bash
# round 5, call au: the poisoned-replay tests six times over, assertion text kept
set -u
O=gpurun_out/r5au
mkdir -p $O
export TMPDIR=/tmp
for i in 1 2 3 4 5 6; do
  timeout 30 python -m pytest tests/test_gpu_graph_step.py -q -k "free_device_memory or equals_the_eager" 2>&1 | grep -E "^E  |passed|failed|^FAILED" | cut -c1-1200 | head -8 >> $O/loop.txt
done
echo done > $O/finished
