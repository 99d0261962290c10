#!/bin/bash
# round 4, call v: same-box A/B of the attention instruction diet at the headline shape: the library of commit 8579e45
# (tools/probes/ab/liblavila_hip_base.so, built beside the tree) against the current one -- space-attention probes at the
# TSF-B shape and the default bench; then the attention tests on the current library
set -u
O=gpurun_out/r4v
mkdir -p $O
export TMPDIR=/tmp
L=lavila_amd/lib/liblavila_hip.so
cp $L /tmp/new.so
for round in 1 2; do
  for v in base new; do
    if [ $v = base ]; then cp tools/probes/ab/liblavila_hip_base.so $L; else cp /tmp/new.so $L; fi
    for m in fwd bwd; do
      echo "$v $m $(timeout 120 python tools/probe_attn.py space $m 256 50 2>&1 | tail -1)" >> $O/probe_ab.txt
    done
  done
done
for v in base new base new; do
  if [ $v = base ]; then cp tools/probes/ab/liblavila_hip_base.so $L; else cp /tmp/new.so $L; fi
  echo "$v $(timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-events 2>/dev/null | grep '^{' | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')" >> $O/bench_ab.txt
done
cp /tmp/new.so $L
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity_bf16.py tests/test_gpu_f32_class.py tests/test_gpu_stream_attention.py -x -q > $O/pytest_attn.log 2>&1
echo "rc=$?" >> $O/pytest_attn.log
for m in fwd bwd; do
  PROBE_F=16 PROBE_N=576 PROBE_H=16 timeout 120 python tools/probe_attn.py space $m 8 30 2>&1 | tail -1 >> $O/probe_config4.txt
done
echo done > $O/finished
