#!/bin/bash
# round 6, call ah: tile timelines of lvl_linear_tn per epilogue (GM_TRACE build): main loop and epilogue durations
set -u
O=gpurun_out/r6ah
mkdir -p $O
export TMPDIR=/tmp
for cfg in "proj 0" "proj 3" "fc1 0" "fc1 1" "fc1 4" "fc1 5" "fc2 3"; do
  set -- $cfg
  echo "=== $1 epilogue $2" >> $O/gemm_tile_trace.txt
  timeout 300 python tools/probe_gemm_trace.py $1 $2 2>&1 | grep -v amdgpu.ids | grep -E "main loop|shader clock|kernel span|wg 0 grp|wg 100 grp" | cut -c1-260 >> $O/gemm_tile_trace.txt
done
echo done > $O/finished
