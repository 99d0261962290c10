#!/bin/bash
# round 6, call aj: soak of the dynamic tile schedule after the mailbox change (160 launches per shape, late-workgroup hook on / off,
# every result compared bit for bit with the static schedule); 60-step bench under LAVILA_DYNAMIC_TILES=1
set -u
O=gpurun_out/r6aj
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python tools/soak_dynamic_tiles.py 2>&1 | grep -v amdgpu.ids > $O/soak.txt
LAVILA_DYNAMIC_TILES=1 timeout 600 python bench.py --steps 60 --warmup 3 --no-cpu-baseline --no-events 2>/dev/null | grep '^{' | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("dynamic 60 steps", d["value"], d["ms_per_step"], d["config"]["final_loss"])' >> $O/soak.txt
timeout 600 python bench.py --steps 60 --warmup 3 --no-cpu-baseline --no-events 2>/dev/null | grep '^{' | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("static 60 steps", d["value"], d["ms_per_step"], d["config"]["final_loss"])' >> $O/soak.txt
echo done > $O/finished
