#!/bin/bash
# round 5, call aa: is the replay's sensitivity tied to the library GEMMs the tiny test geometry falls back to?
set -u
O=gpurun_out/r5aa
mkdir -p $O
export TMPDIR=/tmp
PROBE_COUNT_GEMM=1 PROBE_POISON_ITS=2,3,4 timeout 250 python tools/probe_graph_step_poison.py 2>&1 | grep -v amdgpu.ids | cut -c1-1500 > $O/tiny_count.txt
PROBE_CFG=tsfb PROBE_COUNT_GEMM=1 PROBE_POISON_ITS=2,3,4 timeout 300 python tools/probe_graph_step_poison.py 2>&1 | grep -v amdgpu.ids | cut -c1-1500 > $O/tsfb_count.txt
PROBE_CFG=tsfb timeout 300 python tools/probe_graph_step_poison.py 2>&1 | grep -v amdgpu.ids | cut -c1-1500 > $O/tsfb_all.txt
echo done > $O/finished
