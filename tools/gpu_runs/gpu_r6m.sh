#!/bin/bash
# round 6, call m: GEMM epilogues in scalar f32 (no v_pk_*), per-lane row offsets computed per tile (no hoisting: 248-252 VGPRs,
# no spill in any epilogue): tests, per-shape A/B in random order, same-box bench A/B against the library of call j
set -u
O=gpurun_out/r6m
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity_bf16.py -q -x -k "linear_tn or dynamic_tile or persistent" 2>&1 | grep -E "^E  |passed|failed|^FAILED" | cut -c1-600 | head -20 > $O/tests.txt
timeout 900 python tools/probe_gemm_epilogues.py 2>&1 | grep -v amdgpu.ids | cut -c1-900 > $O/epilogues.txt
AB_BASE_ENV="LAVILA_GELU_DERIV=0" tools/ab_library_swap.sh run $O/ab.txt --steps 10 --warmup 3
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_kernels.py tests/test_gpu_f32_class.py tests/test_gpu_parity_bf16.py -q -x 2>&1 | grep -E "^E  |passed|failed|^FAILED" | cut -c1-600 | head -20 > $O/tests2.txt
echo done > $O/finished
