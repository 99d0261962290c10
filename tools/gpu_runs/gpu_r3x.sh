#!/bin/bash
set -u
O=gpurun_out/r3x
mkdir -p $O
export TMPDIR=/tmp

(timeout 900 python -m pytest tests/test_gpu_narrator.py -m gpu -q 2>&1 | tail -30 | cut -c1-400) > $O/pytest_narrator.log
(timeout 900 python tools/probe_narrator.py --batch 64 --length 77 --returns 10 --sample --half --reps 2 --skip-recompute --out $O/narrator_b64_r10.json 2>&1 | tail -5) > $O/probe_r10.log
echo done > $O/finished
(timeout 900 python tools/probe_narrator.py --batch 64 --length 77 --half --reps 3 --skip-recompute --out $O/narrator_b64.json 2>&1 | tail -5) > $O/probe_b64.log
echo done2 > $O/finished
