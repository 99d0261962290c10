#!/bin/bash
set -u
O=gpurun_out/r3ag
mkdir -p $O
export TMPDIR=/tmp
(timeout 1500 python -m pytest tests/test_gpu_boundary.py tests/test_gpu_narrator.py tests/test_gpu_model.py -m gpu -q 2>&1 | grep -E "^E  |passed|failed|^FAILED|error" | cut -c1-300 | head -40) > $O/pytest.log
(timeout 600 python bench.py --workload narrator --steps 4 --warmup 1 --no-cpu-baseline 2>$O/narr.err | tail -1) > $O/bench_narrator_r10.json
echo done > $O/finished
