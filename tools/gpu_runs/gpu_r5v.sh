#!/bin/bash
# round 5, call v: poison probe variants -- foreach (unfused) AdamW, replay on the capture stream, three repeats of the base
set -u
O=gpurun_out/r5v
mkdir -p $O
export TMPDIR=/tmp
export PROBE_GROUP=0 PROBE_POISON_STREAMS=cur PROBE_POISON_BYTES=256,1048576
r() { name=$1; shift; for i in 1 2 3; do echo "$name run $i: $(env "$@" timeout 200 python tools/probe_graph_step_poison.py 2>&1 | grep -E 'poisoned|max \|dp' | tr '\n' ' ' | cut -c1-260)" >> $O/variants.txt; done; }
r base X=1
r foreach_adamw PROBE_FUSED=0
r replay_on_step_stream PROBE_REPLAY_ON_STEP_STREAM=1
echo done > $O/finished
