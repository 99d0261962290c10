#!/bin/bash
# round 4, call s: streaming forward after the instruction diet (linear LDS-DMA fills without vector address arithmetic,
# branch-free full chunks, three-way maxima without canonicalisation, packed multiply-add / add in the softmax) and its
# 4-workgroup cut with a two-stage ring (variant bit 0)
set -u
O=gpurun_out/r4s
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_stream_attention.py -x -q > $O/pytest_stream.log 2>&1
echo "rc=$?" >> $O/pytest_stream.log
for v in 0 1; do
  PROBE_STREAM_VARIANT=$v PROBE_F=16 PROBE_N=576 PROBE_H=16 timeout 120 python tools/probe_attn.py space fwd 8 30 > $O/probe_fwd_v$v.log 2>&1
done
echo done > $O/finished
