#!/bin/bash
# round 5, call z: NaN fence over every kernel call of one eager iteration
set -u
O=gpurun_out/r5z
mkdir -p $O
export TMPDIR=/tmp
timeout 400 python tools/probe_nan_fence.py tiny 2>&1 | grep -v amdgpu.ids | cut -c1-300 > $O/fence_tiny.txt
timeout 400 python tools/probe_nan_fence.py medium 2>&1 | grep -v amdgpu.ids | cut -c1-300 > $O/fence_medium.txt
echo done > $O/finished
