#!/bin/bash
# round 5, call ah: which gradients of the poisoned replay are non-finite (graph without the optimizer)
set -u
O=gpurun_out/r5ah
mkdir -p $O
export TMPDIR=/tmp
PROBE_VARIANT=nostep PROBE_POISON_ITS=2,3,4 PROBE_POISON_STREAMS=cur PROBE_FILL_SET=0:12 timeout 200 python tools/probe_graph_step_poison.py 2>&1 | grep -v amdgpu.ids | tail -4 | cut -c1-3000 > $O/nostep_grads.txt
PROBE_VARIANT=nostep PROBE_POISON_ITS=4 PROBE_POISON_STREAMS=cur PROBE_FILL_SET=1:2 timeout 200 python tools/probe_graph_step_poison.py 2>&1 | grep -v amdgpu.ids | tail -4 | cut -c1-3000 > $O/nostep_grads_one.txt
echo done > $O/finished
