#!/bin/bash
# round 4, call aa: residual epilogue with two row groups of residual rows in flight (24 B of spills) against one
# (no spills), same box, library swap; GEMM tests on the new library
set -u
O=gpurun_out/r4aa
mkdir -p $O
export TMPDIR=/tmp
L=lavila_amd/lib/liblavila_hip.so
cp $L /tmp/new.so
for v in base new base new; do
  if [ $v = base ]; then cp tools/probes/ab/liblavila_hip_base.so $L; else cp /tmp/new.so $L; fi
  echo "$v $(timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-events 2>/dev/null | grep '^{' | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')" >> $O/bench_ab.txt
done
cp /tmp/new.so $L
timeout 600 python -m pytest tests/test_gpu_parity_bf16.py tests/test_gpu_kernels.py -x -q -k "linear or residual or gemm or mlp" > $O/pytest.log 2>&1
echo "rc=$?" >> $O/pytest.log
echo done > $O/finished
