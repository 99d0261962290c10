#!/bin/bash
# round 5, call at: the two files of call ar again, with the assertion text
set -u
O=gpurun_out/r5at
mkdir -p $O
export TMPDIR=/tmp
timeout 150 python -m pytest tests/test_gpu_ddp.py tests/test_gpu_graph_step.py -q 2>&1 | grep -E "^E  |passed|failed|^FAILED" | cut -c1-900 | head -20 > $O/tests.txt
echo done > $O/finished
