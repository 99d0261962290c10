#!/bin/bash
# round 4, call n: weight gradient on a side stream beside the input gradient (LAVILA_WGRAD_STREAM=1) x tile counters;
# default bench once more for the graphed-step child's idle-device replay time
set -u
O=gpurun_out/r4n
mkdir -p $O
export TMPDIR=/tmp
run() {  # name, env...
  name=$1; shift
  (env "$@" timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-events 2>$O/bench_$name.err | grep '^{' | tail -1) > $O/bench_$name.json
}
run dyn LAVILA_DYNAMIC_TILES=1
run wstream_dyn LAVILA_WGRAD_STREAM=1 LAVILA_DYNAMIC_TILES=1
run wstream_static LAVILA_WGRAD_STREAM=1
run base LAVILA_WGRAD_STREAM=0
timeout 900 python bench.py --steps 8 --warmup 2 --no-cpu-baseline > $O/bench_default.out 2> $O/bench_default.err
echo "rc=$?" >> $O/bench_default.err
echo done > $O/finished
