#!/bin/bash
# round 5, call u: do the poison allocations overlap LIVE tensors (parameters, gradients, optimizer state)?
set -u
O=gpurun_out/r5u
mkdir -p $O
export TMPDIR=/tmp
PROBE_GROUP=0 PROBE_WHERE=1 PROBE_OVERLAP=1 PROBE_POISON_STREAMS=cur,step PROBE_POISON_BYTES=256,1048576,16777216 timeout 300 python tools/probe_graph_step_poison.py 2>&1 | grep -E "^\[it=|^\[poison|^group|max \|dp" > $O/overlap.txt
echo done > $O/finished
