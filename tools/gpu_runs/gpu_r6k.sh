#!/bin/bash
# round 6, call k: GEMM epilogues -- aux rows landed before the first store (residual / multiply-by-aux), packed-f32 QuickGELU,
# the derivative pair (epilogues 4 / 5); per-shape A/B against the round's previous library; start stagger experiment; bench A/B
set -u
O=gpurun_out/r6k
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity_bf16.py -q -x -k "linear_tn or dynamic_tile or persistent" 2>&1 | grep -E "^E  |passed|failed|^FAILED" | cut -c1-600 | head -20 > $O/tests.txt
timeout 900 python tools/probe_gemm_epilogues.py 200 400 800 2>&1 | grep -v amdgpu.ids | cut -c1-900 > $O/epilogues.txt
tools/ab_library_swap.sh run $O/ab.txt --steps 10 --warmup 3
timeout 1200 python -m pytest tests/test_gpu_model.py tests/test_gpu_kernels.py tests/test_gpu_f32_class.py -q -x 2>&1 | grep -E "^E  |passed|failed|^FAILED" | cut -c1-600 | head -20 > $O/tests2.txt
echo done > $O/finished
