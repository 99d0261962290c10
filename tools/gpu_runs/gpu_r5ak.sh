#!/bin/bash
# round 5, call ak: the poison probe after replacing the hipMemsetAsync nodes by a zero-fill kernel
set -u
O=gpurun_out/r5ak
mkdir -p $O
export TMPDIR=/tmp
timeout 200 python tools/probe_graph_step_poison.py 2>&1 | grep -v amdgpu.ids | tail -4 | cut -c1-300 | sed "s/^/[tiny] /" >> $O/after_fix.txt
PROBE_CFG=tsfb timeout 200 python tools/probe_graph_step_poison.py 2>&1 | grep -v amdgpu.ids | tail -4 | cut -c1-300 | sed "s/^/[tsfb] /" >> $O/after_fix.txt
PROBE_GROUP=nccl timeout 200 python tools/probe_graph_step_poison.py 2>&1 | grep -v amdgpu.ids | tail -4 | cut -c1-300 | sed "s/^/[tiny, one-rank RCCL group] /" >> $O/after_fix.txt
for i in 1 2 3 4; do timeout 300 python -m pytest tests/test_gpu_ddp.py -x -q -k "graphed_step_with_a_process_group" 2>&1 | tail -2 | sed "s/^/[2-rank test, interleaved order, run $i] /" >> $O/after_fix.txt; done
echo done > $O/finished
