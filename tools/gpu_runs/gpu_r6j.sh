#!/bin/bash
# round 6, call j: the NaN fence (every C-ABI call of an eager iteration with NaN in all free memory right in front of it, every
# gradient compared with the clean run) at the TSF-B geometries -- VERDICT r5 item 1 asked for it at this geometry
set -u
O=gpurun_out/r6j
mkdir -p $O
export TMPDIR=/tmp
for g in tsfb4 tsfb4x2; do
  timeout 900 python tools/probe_nan_fence.py $g 2>&1 | grep -v amdgpu.ids | tail -12 | cut -c1-300 > $O/fence_$g.txt
  FENCE_GRAPH_PATHS=1 timeout 900 python tools/probe_nan_fence.py $g 2>&1 | grep -v amdgpu.ids | tail -6 | cut -c1-300 > $O/fence_${g}_graph_paths.txt
done
LAVILA_TEXT_STREAM=1 timeout 900 python tools/probe_nan_fence.py tsfb4 2>&1 | grep -v amdgpu.ids | tail -6 | cut -c1-300 > $O/fence_tsfb4_two_streams.txt
echo done > $O/finished
