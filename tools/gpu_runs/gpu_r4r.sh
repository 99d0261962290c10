#!/bin/bash
# round 4, call r: validation of the tree after the streaming DMA rings and the graphed step -- smoke, full GPU suite,
# default bench (with cpu baseline and the graphed-step child)
set -u
O=gpurun_out/r4r
mkdir -p $O
export TMPDIR=/tmp
(timeout 300 python __graft_entry__.py smoke 2>&1 | tail -4) > $O/smoke.log
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -E "^E  |passed|failed|^FAILED|error" | cut -c1-400 | head -40) > $O/pytest.log
(time timeout 900 python bench.py 2>$O/bench.err | grep '^{' | tail -1) > $O/bench.json 2> $O/bench.time
echo done > $O/finished
