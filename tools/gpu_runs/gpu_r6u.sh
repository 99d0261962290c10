#!/bin/bash
# round 6, call u: is the second result tensor of epilogue 4 paid in BYTES or in store instructions? (aux_out = y: same stores, no extra HBM bytes)
set -u
O=gpurun_out/r6u
mkdir -p $O
export TMPDIR=/tmp
PROBE_EPI=4 timeout 600 python tools/probe_gemm_variants.py exp4 2>&1 | grep -v amdgpu.ids > $O/epi4_aux_is_y.txt
PROBE_EPI=4 PROBE_AUX_IS_Y=1 timeout 600 python tools/probe_gemm_variants.py exp4 2>&1 | grep -v amdgpu.ids >> $O/epi4_aux_is_y.txt
echo done > $O/finished
