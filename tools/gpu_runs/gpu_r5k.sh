#!/bin/bash
# round 5, call k: diagnosing the intermittent parameter drift of the 2-rank graphed-step test (verbose, with and without
# the communication stream)
set -u
O=gpurun_out/r5k
mkdir -p $O
export TMPDIR=/tmp
export LAVILA_TEST_VERBOSE=1
for i in 1 2 3 4; do
  (timeout 600 python -m pytest tests/test_gpu_ddp.py -m gpu -q -s -k "two_ranks and graphed" 2>&1 | grep -E "rank [01]\] step|passed|failed" | cut -c1-300) > $O/comm_on_$i.log
done
export LAVILA_GRAPH_COMM_STREAM=0
for i in 1 2 3 4; do
  (timeout 600 python -m pytest tests/test_gpu_ddp.py -m gpu -q -s -k "two_ranks and graphed" 2>&1 | grep -E "rank [01]\] step|passed|failed" | cut -c1-300) > $O/comm_off_$i.log
done
echo done > $O/finished
