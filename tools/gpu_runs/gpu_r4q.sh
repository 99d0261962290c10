#!/bin/bash
# round 4, call q: MFMA loop ceiling of a 4-wave layout (one wave per SIMD, 128x128 per wave: 0.5 fragment reads per MFMA,
# half the barriers) beside the GEMM's 8-wave layout -- is a main-loop rewrite worth it?
set -u
O=gpurun_out/r4q
mkdir -p $O
timeout 300 tools/probes/mfma_ceiling > $O/mfma_ceiling.txt 2>&1
echo "rc=$?" >> $O/mfma_ceiling.txt
echo done > $O/finished
