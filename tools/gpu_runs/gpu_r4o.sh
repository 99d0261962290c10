#!/bin/bash
# round 4, call o: DDP-wrapped model inside the captured iteration (one-rank RCCL group)
set -u
O=gpurun_out/r4o
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_graph_step.py -x -q > $O/pytest_graph.log 2>&1
echo "rc=$?" >> $O/pytest_graph.log
echo done > $O/finished
