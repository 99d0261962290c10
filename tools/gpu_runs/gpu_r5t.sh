#!/bin/bash
# round 5, call t: where does the poison enter (gradients / optimizer state / parameters, which tensors)
set -u
O=gpurun_out/r5t
mkdir -p $O
export TMPDIR=/tmp
PROBE_GROUP=0 PROBE_WHERE=1 PROBE_POISON_STREAMS=cur PROBE_POISON_BYTES=256 timeout 300 python tools/probe_graph_step_poison.py 2>&1 | grep -E "^\[it=|^group|max \|dp" > $O/where_cur256.txt
PROBE_GROUP=0 PROBE_WHERE=1 PROBE_POISON_STREAMS=step PROBE_POISON_BYTES=1048576,16777216 timeout 300 python tools/probe_graph_step_poison.py 2>&1 | grep -E "^\[it=|^group|max \|dp" > $O/where_step_big.txt
echo done > $O/finished
