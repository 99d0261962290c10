#!/bin/bash
# round 5, call m: second bisect of the intermittent 2-rank graphed-step drift (device-wide syncs in the replay chain,
# per-tensor all-reduce, eager and graphed phases separated)
set -u
O=gpurun_out/r5m
mkdir -p $O
export TMPDIR=/tmp
export LAVILA_TEST_VERBOSE=1
t() { name=$1; shift; for i in 1 2 3; do (env "$@" timeout 600 python -m pytest tests/test_gpu_ddp.py -m gpu -q -s -k "two_ranks and graphed" 2>&1 | grep -E "rank 0\] (step [23]|END)|passed|failed" | cut -c1-400) > $O/${name}_$i.log; done; }
t base X=1
t dbg_sync LAVILA_GRAPH_DEBUG_SYNC=1
t pertensor LAVILA_GRAPH_REDUCE=pertensor
t phases LAVILA_TEST_PHASES=1
t phases_sync LAVILA_TEST_PHASES=1 LAVILA_GRAPH_DEBUG_SYNC=1
for f in $O/*.log; do echo "== $f"; cat $f; done > $O/summary.txt
echo done > $O/finished
