#!/bin/bash
# round-3 GPU pass C: tests, bench A/B (text stream on/off), fp8 QK^T rate probe, config-4 attention timing
set -u
O=gpurun_out/r3c
mkdir -p $O
export TMPDIR=/tmp
(timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -40) > $O/pytest.log
timeout 300 python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err
LAVILA_TEXT_STREAM=0 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_no_text_stream.json 2>/dev/null
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_repeat.json 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w tools/probes/fp8_qk_rate.hip -o /tmp/fp8_qk_rate > $O/fp8_build.log 2>&1 && timeout 120 /tmp/fp8_qk_rate > $O/fp8_qk_rate.txt 2>&1
for what in fwd bwd; do
  PROBE_F=16 PROBE_N=576 PROBE_H=16 timeout 200 python tools/probe_attn.py space $what 8 10 >> $O/attn_config4.txt 2>&1
  PROBE_F=16 PROBE_N=576 PROBE_H=16 timeout 200 python tools/probe_attn.py time $what 8 10 >> $O/attn_config4.txt 2>&1
done
timeout 200 python tools/probe_wgrad_mfma.py 8192 tqkv tproj tfc1 tfc2 > $O/wgrad_text.txt 2>&1
echo done > $O/finished
