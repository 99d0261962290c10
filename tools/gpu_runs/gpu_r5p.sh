#!/bin/bash
# round 5, call p: allocation audit of the segmented capture (which allocations of the capture call are NOT in the private pool)
set -u
O=gpurun_out/r5p
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python tools/probe_graph_step_alloc_audit.py > $O/audit_gloo.txt 2>&1
LAVILA_GRAPH_COMM_STREAM=0 timeout 300 python tools/probe_graph_step_alloc_audit.py > $O/audit_gloo_commoff.txt 2>&1
echo done > $O/finished
