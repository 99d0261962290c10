#!/bin/bash
# round 6, call ad: the attention files under other instruction-scheduling strategies (-mllvm -amdgpu-sched-strategy=gcn-max-ilp /
# gcn-iterative-ilp): kernel timing against the tree's build
set -u
O=gpurun_out/r6ad
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python tools/probe_attn_variants.py gcnmaxilp gcniterativeilp 2>&1 | grep -v amdgpu.ids > $O/attn_variants.txt
echo done > $O/finished
