#!/bin/bash
# round 4, call b: f32-class attention (split-operand MFMA kernels) bring-up, new full-size goldens, full suite after the
# attention refactor, rehearsal with the one-stream fix, default bench
set -u
O=gpurun_out/r4b
mkdir -p $O
export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_gpu_f32_class.py -q -s 2>&1 | grep -E "^E  |passed|failed|^FAILED|rror|f32-class" | cut -c1-400 | head -80) > $O/pytest_f32.log
(timeout 1200 python -m pytest tests -m gpu -q --deselect tests/test_gpu_f32_class.py 2>&1 | grep -E "^E  |passed|failed|^FAILED|error" | cut -c1-300 | head -40) > $O/pytest_all.log
(timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1) > $O/bench_1gpu.json
(timeout 400 python bench.py --gpus 2 --batch 32 --steps 6 --warmup 3 --no-cpu-baseline 2>$O/bench_2rank.err | tail -1) > $O/bench_2rank.json
echo done > $O/finished
