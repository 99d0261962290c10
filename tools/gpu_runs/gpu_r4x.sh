#!/bin/bash
# round 4, call x: fp8 QK^T path of the streaming kernels (BASELINE configs[3]): parity test against its emulation, speed
# against the bf16 kernels at the config-4 shape, config-4 bench line with it on; the graph-step test after its fix
set -u
O=gpurun_out/r4x
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_stream_attention.py tests/test_gpu_graph_step.py -x -q > $O/pytest.log 2>&1
echo "rc=$?" >> $O/pytest.log
for fp8 in 0 1; do for m in fwd bwd; do
  echo "fp8=$fp8 $(LAVILA_FP8_QK=$fp8 PROBE_F=16 PROBE_N=576 PROBE_H=16 timeout 120 python tools/probe_attn.py space $m 8 30 2>&1 | tail -1)" >> $O/probe_fp8.txt
done; done
(LAVILA_FP8_QK=1 timeout 600 python bench.py --model CLIP_OPENAI_TIMESFORMER_LARGE_336PX --frames 16 --batch 8 --steps 6 --warmup 2 --no-cpu-baseline 2>$O/bench_config4_fp8.err | grep '^{' | tail -1) > $O/bench_config4_fp8.json
echo done > $O/finished
