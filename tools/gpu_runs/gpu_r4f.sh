#!/bin/bash
# round 4, call f: streaming kernels with branch-free full-chunk bodies: parity + timing (register-budget variants)
set -u
O=gpurun_out/r4f
mkdir -p $O
export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_gpu_stream_attention.py -q 2>&1 | grep -E "^E  |passed|failed|^FAILED|rror" | cut -c1-400 | head -40) > $O/pytest_stream.log
for v in 0 1 2; do for w in fwd bwd; do
  (PROBE_STREAM_VARIANT=$v PROBE_F=16 PROBE_N=576 PROBE_H=16 timeout 300 python tools/probe_attn.py space $w 8 30 2>&1 | tail -1) >> $O/probe_config4_variant$v.txt
done; done
echo done > $O/finished
