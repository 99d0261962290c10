#!/bin/bash
# round 4, call m: default bench with the graphed-step child process
set -u
O=gpurun_out/r4m
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python bench.py --steps 8 --warmup 2 --no-cpu-baseline > $O/bench.out 2> $O/bench.err
echo "rc=$?" >> $O/bench.err
echo done > $O/finished
