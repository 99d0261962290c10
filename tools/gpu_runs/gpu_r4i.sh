#!/bin/bash
# round 4, call i: why did the one-rank RCCL bench not finish within 600 s? per-step progress, three variants, short runs
set -u
O=gpurun_out/r4i
mkdir -p $O
export TMPDIR=/tmp
export LAVILA_BENCH_ONE_RANK_RCCL=1 LAVILA_BENCH_VERBOSE=1
for v in default nostream statictiles; do
  case $v in
    default) E="";;
    nostream) E="LAVILA_TEXT_STREAM=0";;
    statictiles) E="LAVILA_DYNAMIC_TILES=0";;
  esac
  (env $E timeout 150 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-events 2>$O/bench_$v.err | tail -1) > $O/bench_$v.json
  echo "rc=$?" >> $O/bench_$v.err
done
echo done > $O/finished
