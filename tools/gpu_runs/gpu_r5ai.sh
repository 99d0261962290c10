#!/bin/bash
# round 5, call ai: NaN fence over an eager iteration that takes the under-capture branches
set -u
O=gpurun_out/r5ai
mkdir -p $O
export TMPDIR=/tmp
FENCE_GRAPH_PATHS=1 timeout 500 python tools/probe_nan_fence.py tiny 2>&1 | grep -v amdgpu.ids | cut -c1-300 > $O/fence_tiny_graph_paths.txt
echo done > $O/finished
