#!/bin/bash
# round 4, call ai: local-batch sweep of the default workload (what a different per-GPU batch costs or buys)
set -u
O=gpurun_out/r4ai
mkdir -p $O
export TMPDIR=/tmp
for b in 64 128 256 512; do
  echo "batch=$b $(timeout 400 python bench.py --batch $b --steps 8 --warmup 3 --no-cpu-baseline --no-events 2>/dev/null | grep '^{' | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')" >> $O/batch_sweep.txt
done
echo done > $O/finished
