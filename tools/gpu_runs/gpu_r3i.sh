#!/bin/bash
set -u
O=gpurun_out/r3i
mkdir -p $O
export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_gpu_narrator.py -m gpu -q -x 2>&1 | tail -40 | cut -c1-400) > $O/pytest_narrator.log
(timeout 600 python tools/probe_narrator.py --batch 8 --length 20 --half --reps 2 2>&1 | tail -60) > $O/probe_small.log
(timeout 900 python tools/probe_narrator.py --batch 64 --length 77 --half --reps 2 --out $O/narrator_b64.json 2>&1 | tail -60) > $O/probe_b64.log
echo done > $O/finished
