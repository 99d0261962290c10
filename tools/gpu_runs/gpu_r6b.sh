#!/bin/bash
# round 6, call b: cls-gradient partial slots instead of f32 atomics + own text-embedding kernels + node census --
# the GPU suite, the second-outcome probe again (every run must equal the clean one to the bit now), bench
set -u
O=gpurun_out/r6b
mkdir -p $O
export TMPDIR=/tmp
F='amdgpu.ids'
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | cut -c1-400 > $O/tests.txt
timeout 200 python tools/probe_graph_nodes.py 2>&1 | grep -v $F | cut -c1-600 > $O/nodes.txt
for i in 1 2; do
  timeout 300 python tools/probe_second_outcome.py --runs 14 --poison all --fill nan 2>&1 | grep -v $F | cut -c1-300 > $O/all_nan_$i.txt
done
timeout 300 python tools/probe_second_outcome.py --runs 8 --poison all --fill big --calls 5 2>&1 | grep -v $F | cut -c1-300 > $O/all_big.txt
timeout 300 python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err
echo done > $O/finished
