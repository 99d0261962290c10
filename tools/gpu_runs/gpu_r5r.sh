#!/bin/bash
# round 5, call r: which size class / stream of cached free blocks does the replay read? (single graph, no process group)
set -u
O=gpurun_out/r5r
mkdir -p $O
export TMPDIR=/tmp
export PROBE_GROUP=0
for st in cur step; do
  for nb in 256 512 1024 2048 4096 8192 16384 65536 262144 1048576 2097152 4194304 16777216 67108864; do
    echo "stream=$st bytes=$nb $(PROBE_POISON_STREAMS=$st PROBE_POISON_BYTES=$nb timeout 200 python tools/probe_graph_step_poison.py 2>&1 | grep -E 'max \|dp')" >> $O/bisect.txt
  done
done
echo done > $O/finished
