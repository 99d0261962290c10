#!/bin/bash
# round 4, call ah: lean GEMM grid (the smallest grid with the same number of tile rounds: 240 instead of 256 workgroups
# on the N = 768 shapes) against the full grid, same box, library swap; GEMM exactness tests on the new library
set -u
O=gpurun_out/r4ah
mkdir -p $O
export TMPDIR=/tmp
bash tools/ab_library_swap.sh run $O/bench_ab.txt --steps 10 --warmup 3
timeout 600 python -m pytest tests/test_gpu_parity_bf16.py tests/test_gpu_kernels.py tests/test_gpu_f32_class.py -x -q -k "linear or residual or gemm or mlp or persistent" > $O/pytest.log 2>&1
echo "rc=$?" >> $O/pytest.log
echo done > $O/finished
