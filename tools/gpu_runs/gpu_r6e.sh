#!/bin/bash
# round 6, call e: GEMM store-policy A/B (asm stores with their hazard pad), checkpoint test, DDP comm-hook A/B on a one-rank
# RCCL group (default per-parameter division vs the reducer's builtin C++ all-reduce hook)
set -u
O=gpurun_out/r6e
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_model.py -q -x -k "checkpoint" 2>&1 | grep -E "^E  |passed|failed|^FAILED" | cut -c1-600 | head -20 > $O/tests.txt
timeout 600 python tools/probe_gemm_store_policy.py 2>&1 | grep -v amdgpu.ids | cut -c1-400 > $O/store_policy.txt
for v in plain none builtin none builtin; do
  if [ $v = plain ]; then
    r=$(timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-events 2>/dev/null | grep '^{' | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')
  else
    r=$(LAVILA_BENCH_ONE_RANK_RCCL=1 LAVILA_BENCH_DDP_HOOK=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-events 2>/dev/null | grep '^{' | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')
  fi
  echo "$v $r" >> $O/ddp_hook.txt
done
echo done > $O/finished
