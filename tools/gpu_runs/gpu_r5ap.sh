#!/bin/bash
# round 5, call ap: poisoned-replay test incl. the geometry whose embedding backward sorts (rocPRIM memsets inside the graph)
set -u
O=gpurun_out/r5ap
mkdir -p $O
export TMPDIR=/tmp
timeout 240 python -m pytest tests/test_gpu_graph_step.py -q -k "free_device_memory" 2>&1 | tail -25 | cut -c1-400 > $O/poison_tests.txt
echo done > $O/finished
