#!/bin/bash
# round 6, call a: the second outcome of the TSF-B-geometry step under a poisoned allocator -- attribution runs,
# graph node census, baseline bench of the round's starting tree
set -u
O=gpurun_out/r6a
mkdir -p $O
export TMPDIR=/tmp
F='amdgpu.ids'
timeout 200 python tools/probe_graph_nodes.py 2>&1 | grep -v $F | cut -c1-600 > $O/nodes.txt
for i in 1 2 3; do
  timeout 200 python tools/probe_second_outcome.py --runs 6 --poison all --fill nan 2>&1 | grep -v $F | cut -c1-400 > $O/all_nan_$i.txt
done
timeout 200 python tools/probe_second_outcome.py --runs 6 --poison all --fill nan --text-stream 0 2>&1 | grep -v $F | cut -c1-400 > $O/all_nan_ts0.txt
timeout 200 python tools/probe_second_outcome.py --runs 6 --poison all --fill big 2>&1 | grep -v $F | cut -c1-400 > $O/all_big.txt
timeout 200 python tools/probe_second_outcome.py --runs 6 --poison all --fill nan --eager-only 2>&1 | grep -v $F | cut -c1-400 > $O/eager_only_nan.txt
timeout 200 python tools/probe_second_outcome.py --runs 6 --poison eager --fill nan 2>&1 | grep -v $F | cut -c1-400 > $O/eager_nan.txt
timeout 300 python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err
echo done > $O/finished
