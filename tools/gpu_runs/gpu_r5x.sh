#!/bin/bash
# round 5, call x: which poison matters -- the one in front of the capture call or the ones in front of the replays?
set -u
O=gpurun_out/r5x
mkdir -p $O
export TMPDIR=/tmp
for its in 0 1 2 3 0,1 2,3,4; do
  PROBE_POISON_ITS=$its timeout 200 python tools/probe_graph_step_poison.py 2>&1 | grep -v amdgpu.ids | tail -3 | sed "s/^/[its=$its] /" >> $O/which_poison.txt
done
PROBE_POISON_ITS=2,3,4 PROBE_WHERE=1 timeout 200 python tools/probe_graph_step_poison.py 2>&1 | grep -v amdgpu.ids | cut -c1-3000 > $O/where_234.txt
PROBE_POISON_ITS=0,1 PROBE_WHERE=1 timeout 200 python tools/probe_graph_step_poison.py 2>&1 | grep -v amdgpu.ids | cut -c1-3000 > $O/where_01.txt
echo done > $O/finished
