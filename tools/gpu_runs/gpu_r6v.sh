#!/bin/bash
# round 6, call v: store burst of a 256x256 bf16 tile from every CU, quarter-line stores (the GEMM epilogue) vs full lines, streaming
# (HBM-bound) and L2-resident (what the L2 accepts)
set -u
O=gpurun_out/r6v
mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/probes/store_burst.hip -o /tmp/store_burst 2>/dev/null && timeout 120 /tmp/store_burst > $O/store_burst.txt 2>&1
echo done > $O/finished
