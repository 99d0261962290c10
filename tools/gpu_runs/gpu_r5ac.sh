#!/bin/bash
# round 5, call ac: which tensors of the poisoned replay go wrong first (gradients / optimizer state / parameters)?
set -u
O=gpurun_out/r5ac
mkdir -p $O
export TMPDIR=/tmp
PROBE_SAVE=1 PROBE_POISON_ITS=2 timeout 300 python tools/probe_graph_step_poison.py 2>&1 | grep -v amdgpu.ids | cut -c1-1800 > $O/save_its2.txt
PROBE_SAVE=1 PROBE_POISON_ITS=2 PROBE_FUSED=0 timeout 300 python tools/probe_graph_step_poison.py 2>&1 | grep -v amdgpu.ids | cut -c1-1800 > $O/save_its2_foreach.txt
echo done > $O/finished
