#!/bin/bash
# round 5, call af: the live tensors in front of the culprit poison tensor
set -u
O=gpurun_out/r5af
mkdir -p $O
export TMPDIR=/tmp
PROBE_POISON_ITS=2 PROBE_POISON_STREAMS=cur PROBE_FILL_SET=1:2 PROBE_DESCRIBE=1 timeout 300 python tools/probe_graph_step_poison.py 2>&1 | grep -v amdgpu.ids | grep -v "listcomp" | cut -c1-300 > $O/describe.txt
echo done > $O/finished
