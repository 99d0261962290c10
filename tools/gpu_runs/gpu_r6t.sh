#!/bin/bash
# round 6, call t: where the QuickGELU + derivative epilogue's time goes (GM_EXP builds: no second store / no arithmetic / neither)
set -u
O=gpurun_out/r6t
mkdir -p $O
export TMPDIR=/tmp
PROBE_EPI=4 timeout 600 python tools/probe_gemm_variants.py exp4 exp8 exp12 2>&1 | grep -v amdgpu.ids > $O/epi4_variants.txt
PROBE_EPI=0 timeout 600 python tools/probe_gemm_variants.py 2>&1 | grep -v amdgpu.ids | grep fc1 >> $O/epi4_variants.txt
echo done > $O/finished
