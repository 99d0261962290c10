#!/bin/bash
# round 5, call l: bisect of the 2-rank graphed-step drift by switch
set -u
O=gpurun_out/r5l
mkdir -p $O
export TMPDIR=/tmp
export LAVILA_TEST_VERBOSE=1
t() { name=$1; shift; for i in 1 2; do (env "$@" timeout 600 python -m pytest tests/test_gpu_ddp.py -m gpu -q -s -k "two_ranks and graphed" 2>&1 | grep -E "rank 0\] step [23]|passed|failed" | cut -c1-300) > $O/${name}_$i.log; done; }
t base X=1
t embed_off LAVILA_EMBED_BWD_KERNEL=0
t tiles_static LAVILA_DYNAMIC_TILES=0
t tokens_off LAVILA_COLSUM_TOKENS=0
t rider0 LAVILA_TIME_BWD_RIDER=0
t residual_off LAVILA_RESIDUAL_EPILOGUE=0
t cls_last_off LAVILA_CLS_LAST=0
for f in $O/*.log; do echo "== $f"; cat $f; done > $O/summary.txt
echo done > $O/finished
