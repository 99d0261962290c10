#!/bin/bash
# round 5, call q: poison probe of GraphedTrainStep with and without a process group
set -u
O=gpurun_out/r5q
mkdir -p $O
export TMPDIR=/tmp
for g in 0 gloo nccl; do PROBE_GROUP=$g timeout 300 python tools/probe_graph_step_poison.py 2>&1 | grep -E "^group|max \|dp" > $O/poison_$g.txt; done
LAVILA_COLSUM_TOKENS=0 PROBE_GROUP=0 timeout 300 python tools/probe_graph_step_poison.py 2>&1 | grep -E "^group|max \|dp" > $O/poison_0_tokens_off.txt
LAVILA_EMBED_BWD_KERNEL=0 LAVILA_COLSUM_TOKENS=0 LAVILA_TIME_BWD_RIDER=0 PROBE_GROUP=0 timeout 300 python tools/probe_graph_step_poison.py 2>&1 | grep -E "^group|max \|dp" > $O/poison_0_all_off.txt
(cd _r4_tree && PROBE_GROUP=0 timeout 300 python ../tools/probe_graph_step_poison.py 2>&1 | grep -E "^group|max \|dp|Error" | tail -4) > $O/poison_r4_tree.txt
for f in $O/*.txt; do echo "== $f"; cat $f; done > $O/summary.log
echo done > $O/finished
