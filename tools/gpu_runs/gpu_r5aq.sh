#!/bin/bash
# round 5, call aq: tile timelines of the TN GEMM on the round's tree (GM_TRACE build of the unchanged kernel)
set -u
O=gpurun_out/r5aq
mkdir -p $O
export TMPDIR=/tmp
for s in qkv proj fc1 fc2 dqkv; do
  echo "=== $s" >> $O/gemm_tile_trace.txt
  timeout 60 python tools/probe_gemm_trace.py $s 2>&1 | grep -v amdgpu.ids | grep -v "cycles:" | cut -c1-300 >> $O/gemm_tile_trace.txt
done
echo done > $O/finished
