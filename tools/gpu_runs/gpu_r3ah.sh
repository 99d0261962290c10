#!/bin/bash
set -u
O=gpurun_out/r3ah
mkdir -p $O
export TMPDIR=/tmp
for r in 1024 2048 8192 65536; do
  (LAVILA_SKINNY_MAX_ROWS=$r timeout 600 python tools/probe_narrator.py --batch 64 --length 77 --half --reps 1 --modes recompute --out $O/recompute_$r.json 2>&1 | tail -2) > $O/log_$r.txt
done
echo done > $O/finished
