#!/bin/bash
# round 5, call ae: bisect the poison tensors, describe the memory around the culprit
set -u
O=gpurun_out/r5ae
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python tools/probe_graph_step_bisect.py > $O/bisect.txt 2>&1
echo done > $O/finished
