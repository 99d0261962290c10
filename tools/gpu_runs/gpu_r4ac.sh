#!/bin/bash
# round 4, call ac: PMC traffic passes of the bench step on the final tree (the GEMM mix now carries the residual reads)
set -u
O=gpurun_out/r4ac
mkdir -p $O
export TMPDIR=/tmp
bash tools/pmc_bench_traffic.sh > $O/traffic.log 2>&1
cp gpurun_out/traffic/*.json $O/ 2>/dev/null
rm -rf gpurun_out/traffic/FETCH_SIZE gpurun_out/traffic/WRITE_SIZE
echo done > $O/finished
