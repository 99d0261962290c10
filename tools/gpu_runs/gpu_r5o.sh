#!/bin/bash
# round 5, call o: poison test -- phases order (no eager iteration between replays) + NaN fill of all cached free blocks
# between replays: does the graph chain read memory it does not own?
set -u
O=gpurun_out/r5o
mkdir -p $O
export TMPDIR=/tmp
export LAVILA_TEST_VERBOSE=1
t() { name=$1; shift; for i in 1 2 3; do (env "$@" timeout 600 python -m pytest tests/test_gpu_ddp.py -m gpu -q -s -k "$K" 2>&1 | grep -E "rank 0\] END|passed|failed" | cut -c1-420) > $O/${name}_$i.log; done; }
K="two_ranks and graphed"
t w2_phases_poison LAVILA_TEST_PHASES=1 LAVILA_TEST_POISON=1
t w2_phases_poison_commoff LAVILA_TEST_PHASES=1 LAVILA_TEST_POISON=1 LAVILA_GRAPH_COMM_STREAM=0
K="live_rccl"
t w1_poison LAVILA_TEST_POISON=1
for f in $O/*.log; do echo "== $f"; cat $f; done > $O/summary.txt
echo done > $O/finished
