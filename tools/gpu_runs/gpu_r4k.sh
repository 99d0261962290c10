#!/bin/bash
# round 4, call k: the pretraining iteration as one hipGraph (lavila_amd/graph_step.py): tests, bench line with the
# graphed_step variant
set -u
O=gpurun_out/r4k
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_graph_step.py -x -q > $O/pytest_graph.log 2>&1
echo "rc=$?" >> $O/pytest_graph.log
timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline > $O/bench.out 2> $O/bench.err
echo "rc=$?" >> $O/bench.err
echo done > $O/finished
