#!/bin/bash
# round 5, call s: where do the poison allocations land (private-pool segments?), single graph, no process group
set -u
O=gpurun_out/r5s
mkdir -p $O
export TMPDIR=/tmp
PROBE_GROUP=0 PROBE_CLASSIFY=1 PROBE_POISON_STREAMS=cur PROBE_POISON_BYTES=256,1048576,16777216 timeout 300 python tools/probe_graph_step_poison.py 2>&1 | grep -E "poison it|^group|max \|dp" > $O/classify_cur.txt
PROBE_GROUP=0 PROBE_CLASSIFY=1 PROBE_POISON_STREAMS=step PROBE_POISON_BYTES=256,1048576,16777216 timeout 300 python tools/probe_graph_step_poison.py 2>&1 | grep -E "poison it|^group|max \|dp" > $O/classify_step.txt
echo done > $O/finished
