#!/bin/bash
# round 5, call y: previous owners of the memory whose poisoning breaks the replay
set -u
O=gpurun_out/r5y
mkdir -p $O
export TMPDIR=/tmp
PROBE_POISON_ITS=2 PROBE_POISON_STREAMS=cur PROBE_POISON_BYTES=256 PROBE_OWNERS=1 timeout 250 python tools/probe_graph_step_poison.py 2>&1 | grep -v amdgpu.ids | cut -c1-700 > $O/owners_256.txt
PROBE_POISON_ITS=2 PROBE_POISON_STREAMS=cur PROBE_POISON_BYTES=4194304 PROBE_OWNERS=1 timeout 250 python tools/probe_graph_step_poison.py 2>&1 | grep -v amdgpu.ids | cut -c1-700 > $O/owners_4m.txt
PROBE_POISON_ITS=2 PROBE_POISON_STREAMS=step PROBE_POISON_BYTES=256 PROBE_OWNERS=1 timeout 250 python tools/probe_graph_step_poison.py 2>&1 | grep -v amdgpu.ids | cut -c1-700 > $O/owners_step_256.txt
echo done > $O/finished
