#!/bin/bash
# round 6, call ac: bench A/B LAVILA_WEIGHT_REFRESH=0 / 1, three alternations
set -u
O=gpurun_out/r6ac
mkdir -p $O
export TMPDIR=/tmp
for e in 1 0 1 0 1 0; do
  LAVILA_WEIGHT_REFRESH=$e timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-events 2>/dev/null | grep '^{' | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("REFRESH='$e'", d["value"], d["ms_per_step"])' >> $O/ab.txt
done
echo done > $O/finished
