#!/bin/bash
# round 4, call c: streaming space-attention kernels (config 4): parity, timing against the resident kernels, config-4 bench line
set -u
O=gpurun_out/r4c
mkdir -p $O
export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_gpu_stream_attention.py tests/test_gpu_f32_class.py -q -x 2>&1 | grep -E "^E  |passed|failed|^FAILED|rror" | cut -c1-400 | head -60) > $O/pytest_stream.log
(timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity_bf16.py -q 2>&1 | grep -E "^E  |passed|failed|^FAILED|rror" | cut -c1-300 | head -40) > $O/pytest_kernels.log
for st in 0 -1; do
  for w in fwd bwd; do
    (PROBE_STREAM=$st PROBE_F=16 PROBE_N=576 PROBE_H=16 timeout 300 python tools/probe_attn.py space $w 8 20 2>&1 | tail -1) >> $O/probe_config4_stream$st.txt
  done
done
# the resident kernels' home turf, for reference: TSF-B shape forced through the streaming kernels
for st in 0 1; do
  for w in fwd bwd; do
    (PROBE_STREAM=$st timeout 300 python tools/probe_attn.py space $w 256 20 2>&1 | tail -1) >> $O/probe_tsfb_stream$st.txt
  done
done
(timeout 900 python bench.py --model CLIP_OPENAI_TIMESFORMER_LARGE_336PX --frames 16 --batch 8 --steps 6 --warmup 2 --no-cpu-baseline 2>$O/bench_config4.err | tail -1) > $O/bench_config4.json
echo done > $O/finished
