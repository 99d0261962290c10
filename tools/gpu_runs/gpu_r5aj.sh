#!/bin/bash
# round 5, call aj: the pattern of the non-finite gradient entries
set -u
O=gpurun_out/r5aj
mkdir -p $O
export TMPDIR=/tmp
PROBE_VARIANT=nostep PROBE_POISON_ITS=2,3,4 PROBE_POISON_STREAMS=cur PROBE_FILL_SET=0:12 timeout 200 python tools/probe_graph_step_poison.py 2>&1 | grep -v amdgpu.ids | grep "^   " | cut -c1-400 > $O/pattern.txt
echo done > $O/finished
