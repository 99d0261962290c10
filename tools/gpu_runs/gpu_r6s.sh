#!/bin/bash
# round 6, call s: tile-queue mailbox read as an LDS instruction (was a flat load + vmcnt(0) drain per tile): per-shape A/B static vs
# dynamic schedule (base = library of call m), dynamic-schedule tests, bench with LAVILA_DYNAMIC_TILES=1 on both libraries
set -u
O=gpurun_out/r6s
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity_bf16.py -q -x -k "dynamic or late or steal or wgrad" 2>&1 | grep -E "^E  |passed|failed|^FAILED" | cut -c1-400 | head > $O/tests.txt
timeout 900 python tools/probe_gemm_epilogues.py 2>&1 | grep -v amdgpu.ids | cut -c1-900 > $O/epilogues.txt
export LAVILA_DYNAMIC_TILES=1
AB_BASE_ENV="LAVILA_GELU_DERIV=0" tools/ab_library_swap.sh run $O/ab_dynamic.txt --steps 10 --warmup 3
unset LAVILA_DYNAMIC_TILES
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-events 2>/dev/null | grep '^{' | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("new_static", d["value"], d["ms_per_step"])' >> $O/ab_dynamic.txt
echo done > $O/finished
