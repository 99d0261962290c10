#!/bin/bash
# round 6, last check of the committed tree: GPU suite, smoke, default bench line
set -u
O=gpurun_out/r6final2
mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | grep -E "^E  |passed|failed|^FAILED" | cut -c1-300 | head -20 > $O/tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -6 | cut -c1-400 > $O/smoke.txt
(timeout 600 python bench.py 2>$O/bench.err | grep '^{' | tail -1) > $O/bench.json
echo done > $O/finished
