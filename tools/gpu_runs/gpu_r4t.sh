#!/bin/bash
# round 4, call t: streaming kernels after the instruction diet (forward: 4-workgroup cut by default; dq / dkv: linear
# fills): tests, probes, PMC passes of the forward, config-4 bench line
set -u
O=gpurun_out/r4t
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_stream_attention.py -x -q > $O/pytest_stream.log 2>&1
echo "rc=$?" >> $O/pytest_stream.log
for m in fwd bwd; do
  PROBE_F=16 PROBE_N=576 PROBE_H=16 timeout 120 python tools/probe_attn.py space $m 8 30 > $O/probe_$m.log 2>&1
done
PROBE_F=16 PROBE_N=576 PROBE_H=16 timeout 400 bash tools/pmc_probe.sh stream_fwd_r4t space fwd 8 3 > $O/pmc_fwd.log 2>&1
cp gpurun_out/pmc_stream_fwd_r4t/summary.txt $O/pmc_stream_fwd_summary.txt 2>/dev/null
(timeout 600 python bench.py --model CLIP_OPENAI_TIMESFORMER_LARGE_336PX --frames 16 --batch 8 --steps 6 --warmup 2 --no-cpu-baseline 2>$O/bench_config4.err | grep '^{' | tail -1) > $O/bench_config4.json
echo done > $O/finished
