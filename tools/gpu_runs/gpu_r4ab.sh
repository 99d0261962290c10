#!/bin/bash
# round 4, call ab: the default bench line of the final tree (by_epilogue split of the GEMM roofline)
set -u
O=gpurun_out/r4ab
mkdir -p $O
export TMPDIR=/tmp
(time timeout 900 python bench.py 2>$O/bench.err | grep '^{' | tail -1) > $O/bench.json 2> $O/bench.time
echo done > $O/finished
