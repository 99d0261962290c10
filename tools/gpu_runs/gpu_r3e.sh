#!/bin/bash
set -u
O=gpurun_out/r3e
mkdir -p $O
export TMPDIR=/tmp
for s in qkv proj fc2 dqkv; do
  echo "=== $s" >> $O/gemm_trace.txt
  timeout 120 python tools/probe_gemm_trace.py $s 2>&1 | grep -v amdgpu >> $O/gemm_trace.txt
done
echo done > $O/finished
