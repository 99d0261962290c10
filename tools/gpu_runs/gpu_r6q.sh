#!/bin/bash
# round 6, call q: what the K loop's barriers and vmcnt waits cost (GM_EXP builds: timing only); full GPU suite on the tree
set -u
O=gpurun_out/r6q
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python tools/probe_gemm_variants.py exp1 exp2 exp3 2>&1 | grep -v amdgpu.ids > $O/gemm_variants.txt
timeout 2700 python -m pytest tests -q -m gpu 2>&1 | grep -E "^E  |passed|failed|^FAILED" | cut -c1-400 | head -30 > $O/tests.txt
echo done > $O/finished
