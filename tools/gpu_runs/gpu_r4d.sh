#!/bin/bash
# round 4, call d: narrator beam search on the HIP path, streaming kernels with the XCD-aware block decode (tests, timing,
# config-4 bench line), float32 577-key timing (split-operand streaming vs the generic kernels)
set -u
O=gpurun_out/r4d
mkdir -p $O
export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_gpu_narrator.py -q -x -k "beam or loud or generate_matches" 2>&1 | grep -E "^E  |passed|failed|^FAILED|rror" | cut -c1-400 | head -60) > $O/pytest_beam.log
(timeout 900 python -m pytest tests/test_gpu_stream_attention.py tests/test_gpu_f32_class.py -q 2>&1 | grep -E "^E  |passed|failed|^FAILED|rror" | cut -c1-400 | head -40) > $O/pytest_stream.log
for w in fwd bwd; do
  (PROBE_F=16 PROBE_N=576 PROBE_H=16 timeout 300 python tools/probe_attn.py space $w 8 20 2>&1 | tail -1) >> $O/probe_config4_stream_xcd.txt
done
(timeout 900 python bench.py --model CLIP_OPENAI_TIMESFORMER_LARGE_336PX --frames 16 --batch 8 --steps 6 --warmup 2 --no-cpu-baseline 2>$O/bench_config4.err | tail -1) > $O/bench_config4.json
(timeout 900 python bench.py --frames 16 --batch 64 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1) > $O/bench_config3_shape.json
echo done > $O/finished
