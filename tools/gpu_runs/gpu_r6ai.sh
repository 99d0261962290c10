#!/bin/bash
# round 6, call ai: is the aux_in epilogue's wait HBM? proj + residual tile trace with the residual tensor cache-resident (small M)
set -u
O=gpurun_out/r6ai
mkdir -p $O
export TMPDIR=/tmp
for m in 200960 65536 21760; do
  for e in 0 3; do
    echo "=== proj epilogue $e M=$m" >> $O/trace_small_m.txt
    PROBE_M=$m timeout 300 python tools/probe_gemm_trace.py proj $e 2>&1 | grep -E "main loop|shader clock|kernel span" | cut -c1-200 >> $O/trace_small_m.txt
  done
done
echo done > $O/finished
