#!/bin/bash
# round 6, call ag: full-size property test of the QuickGELU derivative pair
set -u
O=gpurun_out/r6ag
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity_bf16.py -q -x -k "quickgelu" 2>&1 | grep -E "^E  |passed|failed|^FAILED" | cut -c1-400 | head > $O/tests.txt
echo done > $O/finished
