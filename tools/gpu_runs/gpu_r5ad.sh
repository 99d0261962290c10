#!/bin/bash
# round 5, call ad: guarded allocator (2 MiB of NaN either side of every tensor) under one eager iteration
set -u
O=gpurun_out/r5ad
mkdir -p $O
export TMPDIR=/tmp
for c in tiny medium tsfb; do
  timeout 400 python tools/probe_guard_alloc.py $c 2>&1 | grep -v amdgpu.ids | cut -c1-400 > $O/guard_$c.txt
done
echo done > $O/finished
