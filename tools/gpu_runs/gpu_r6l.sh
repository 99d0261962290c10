#!/bin/bash
# round 6, call l: VALU micro-benchmark of the QuickGELU epilogue forms (packed f32 measured SLOWER in call k); same-box bench A/B
# of the epilogue changes (base = HEAD of call j, pre-activation form of the MLP on both ABI sides)
set -u
O=gpurun_out/r6l
mkdir -p $O
export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/probes/valu_gelu.hip -o /tmp/valu_gelu 2>/dev/null && timeout 120 /tmp/valu_gelu > $O/valu_gelu.txt 2>&1
AB_BASE_ENV="LAVILA_GELU_DERIV=0" tools/ab_library_swap.sh run $O/ab.txt --steps 10 --warmup 3
LAVILA_GELU_DERIV=0 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-events 2>/dev/null | grep '^{' | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("new_lib_preact", d["value"], d["ms_per_step"])' >> $O/ab.txt
echo done > $O/finished
