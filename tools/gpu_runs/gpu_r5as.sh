#!/bin/bash
# round 5, call as: the tsfb poisoned-replay test, three times, with the assertion text
set -u
O=gpurun_out/r5as
mkdir -p $O
export TMPDIR=/tmp
for i in 1 2 3; do
  timeout 100 python -m pytest "tests/test_gpu_graph_step.py::test_replay_does_not_depend_on_free_device_memory[tsfb]" -q 2>&1 | grep -E "^E  |passed|failed" | cut -c1-700 | head -12 >> $O/tsfb.txt
done
echo done > $O/finished
