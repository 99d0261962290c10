#!/bin/bash
# round 4, call a: f32-class GEMM bring-up, full GPU suite, default bench, 2-rank rehearsal bisect
set -u
O=gpurun_out/r4a
mkdir -p $O
export TMPDIR=/tmp
(timeout 600 python -m pytest tests/test_gpu_f32_class.py -q -s 2>&1 | grep -E "^E  |passed|failed|^FAILED|rror|f32-class" | cut -c1-400 | head -60) > $O/pytest_f32.log
(timeout 1200 python -m pytest tests -m gpu -q --deselect tests/test_gpu_f32_class.py 2>&1 | grep -E "^E  |passed|failed|^FAILED|error" | cut -c1-300 | head -40) > $O/pytest_all.log
(timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1) > $O/bench_1gpu.json
for v in default trimoff nostream statictiles; do
  case $v in
    default) E="";;
    trimoff) E="LAVILA_TEXT_TRIM=0";;
    nostream) E="LAVILA_TEXT_STREAM=0";;
    statictiles) E="LAVILA_DYNAMIC_TILES=0";;
  esac
  (env $E timeout 400 python bench.py --gpus 2 --batch 32 --steps 6 --warmup 3 --no-events --no-cpu-baseline 2>$O/bench_2rank_$v.err | tail -1) > $O/bench_2rank_$v.json
done
echo done > $O/finished
