#!/bin/bash
# round 5, call w: free audit around the capture call
set -u
O=gpurun_out/r5w
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python tools/probe_graph_step_free_audit.py > $O/free_audit.txt 2>&1
echo done > $O/finished
