#!/bin/bash
# round 5, call an: event stride in the default bench; kernarg placement A/B (HIP_FORCE_DEV_KERNARG)
set -u
O=gpurun_out/r5an
mkdir -p $O
export TMPDIR=/tmp
LAVILA_BENCH_GRAPH=0 timeout 200 python bench.py --steps 8 --no-cpu-baseline > $O/bench_stride4.json 2> $O/bench_stride4.err
LAVILA_BENCH_GRAPH=0 HIP_FORCE_DEV_KERNARG=1 timeout 200 python bench.py --steps 8 --no-cpu-baseline > $O/bench_devkernarg1.json 2>/dev/null
LAVILA_BENCH_GRAPH=0 HIP_FORCE_DEV_KERNARG=0 timeout 200 python bench.py --steps 8 --no-cpu-baseline > $O/bench_devkernarg0.json 2>/dev/null
LAVILA_BENCH_GRAPH=0 timeout 200 python bench.py --steps 8 --no-cpu-baseline --event-stride 1 > $O/bench_stride1.json 2>/dev/null
echo done > $O/finished
