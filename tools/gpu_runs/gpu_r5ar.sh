#!/bin/bash
# round 5, call ar: the multi-rank GPU tests on the last tree (file touched after the last full-suite run)
set -u
O=gpurun_out/r5ar
mkdir -p $O
export TMPDIR=/tmp
timeout 200 python -m pytest tests/test_gpu_ddp.py tests/test_gpu_graph_step.py -q 2>&1 | tail -6 | cut -c1-300 > $O/tests.txt
echo done > $O/finished
