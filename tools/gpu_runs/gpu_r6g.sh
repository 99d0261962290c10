#!/bin/bash
# round 6, call g: the two new whole-model tests (determinism of the full step, batch-32 bf16 vs f32-class) with their printed
# numbers, then the whole GPU suite
set -u
O=gpurun_out/r6g
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity_bf16.py -q -x -s -k "function_of_its_inputs or batch_32" 2>&1 | grep -E "^\[bf16|^E  |passed|failed|^FAILED" | cut -c1-600 | head -20 > $O/new_tests.txt
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | cut -c1-300 > $O/tests.txt
echo done > $O/finished
