#!/bin/bash
# round 5, call ab: the replay's sensitivity with every GEMM on lavila_amd's own kernels (small-row wgrads padded)
set -u
O=gpurun_out/r5ab
mkdir -p $O
export TMPDIR=/tmp
PROBE_CFG=tiny256 PROBE_COUNT_GEMM=1 timeout 250 python tools/probe_graph_step_poison.py 2>&1 | grep -v amdgpu.ids | cut -c1-1500 > $O/tiny256_count.txt
PROBE_CFG=tiny256 timeout 250 python tools/probe_graph_step_poison.py 2>&1 | grep -v amdgpu.ids | cut -c1-1500 > $O/tiny256.txt
PROBE_CFG=tsfb PROBE_COUNT_GEMM=1 timeout 300 python tools/probe_graph_step_poison.py 2>&1 | grep -v amdgpu.ids | cut -c1-1500 > $O/tsfb_count.txt
PROBE_CFG=tsfb timeout 300 python tools/probe_graph_step_poison.py 2>&1 | grep -v amdgpu.ids | cut -c1-1500 > $O/tsfb.txt
timeout 300 python tools/probe_graph_step_poison.py 2>&1 | grep -v amdgpu.ids | cut -c1-1500 > $O/tiny.txt
echo done > $O/finished
