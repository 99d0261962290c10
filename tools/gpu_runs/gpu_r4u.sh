#!/bin/bash
# round 4, call u: resident space forward kernel with three-way maxima (no canonicalising v_max) and packed score pairs:
# attention tests, probe at the TSF-B shape
set -u
O=gpurun_out/r4u
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity_bf16.py tests/test_gpu_f32_class.py -x -q -k "attention or attn or space or causal or text or one_hot" > $O/pytest_attn.log 2>&1
echo "rc=$?" >> $O/pytest_attn.log
for i in 1 2; do
  timeout 120 python tools/probe_attn.py space fwd 256 50 2>&1 | tail -1 >> $O/probe_fwd.log
done
echo done > $O/finished
