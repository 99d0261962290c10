#!/bin/bash
# round 4, call l: bisect of the whole-iteration capture crash (hipStreamEndCapture segfault in call k)
set -u
O=gpurun_out/r4l
mkdir -p $O
export TMPDIR=/tmp
for cfg in "gstep 1" "gstep+host 1"; do
  set -- $cfg
  timeout 120 python tools/probe_graph_step.py $1 $2 > $O/probe_$1_$2.log 2>&1
  echo "rc=$?" >> $O/probe_$1_$2.log
done
timeout 600 python -m pytest tests/test_gpu_graph_step.py -x -q > $O/pytest_graph.log 2>&1
echo "rc=$?" >> $O/pytest_graph.log
timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline > $O/bench.out 2> $O/bench.err
echo "rc=$?" >> $O/bench.err
echo done > $O/finished
