#!/bin/bash
# round 6, call aa: fused space backward with scalar score arithmetic (no v_pk_fma / v_pk_mul_f32) against the packed form
set -u
O=gpurun_out/r6aa
mkdir -p $O
export TMPDIR=/tmp
for i in 1 2; do timeout 300 python tools/probe_space_bwd_ab.py space 2>&1 | grep -v amdgpu.ids >> $O/attn_ab.txt; done
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity_bf16.py -q -x -k "attention or attn or divided" 2>&1 | grep -E "^E  |passed|failed|^FAILED" | cut -c1-400 | head > $O/tests.txt
echo done > $O/finished
