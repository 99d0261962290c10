#!/bin/bash
# round 5, call n: third bisect of the interleaving-dependent drift (eager model without collectives, separate loss object,
# empty_cache between the eager and the graphed half)
set -u
O=gpurun_out/r5n
mkdir -p $O
export TMPDIR=/tmp
export LAVILA_TEST_VERBOSE=1
t() { name=$1; shift; for i in 1 2 3; do (env "$@" timeout 600 python -m pytest tests/test_gpu_ddp.py -m gpu -q -s -k "two_ranks and graphed" 2>&1 | grep -E "rank 0\] END|passed|failed" | cut -c1-420) > $O/${name}_$i.log; done; }
t eager_no_ddp LAVILA_TEST_EAGER_NO_DDP=1
t separate_crit LAVILA_TEST_SEPARATE_CRIT=1
t empty_cache LAVILA_TEST_EMPTY_CACHE=1
t tokens_tiles_static LAVILA_DYNAMIC_TILES=0 LAVILA_TEST_EAGER_NO_DDP=1
for f in $O/*.log; do echo "== $f"; cat $f; done > $O/summary.txt
echo done > $O/finished
