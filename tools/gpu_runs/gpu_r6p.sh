#!/bin/bash
# round 6, call p: VALU rates incl. v_cndmask forms; bench A/B: library of call m (base) vs per-file SLP decisions
set -u
O=gpurun_out/r6p
mkdir -p $O
export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/probes/valu_rates.hip -o /tmp/valu_rates 2>/dev/null && timeout 300 /tmp/valu_rates > $O/valu_rates.txt 2>&1
tools/ab_library_swap.sh run $O/ab.txt --steps 10 --warmup 3
echo done > $O/finished
