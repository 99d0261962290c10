#!/bin/bash
set -u
O=gpurun_out/r3ab
mkdir -p $O
export TMPDIR=/tmp
(timeout 600 python bench.py --workload narrator --steps 4 --warmup 1 2>$O/narr_r10.err | tail -1) > $O/bench_narrator_r10.json
(timeout 600 python bench.py --workload narrator --returns 1 --steps 4 --warmup 1 2>$O/narr_r1.err | tail -1) > $O/bench_narrator_r1.json
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -E "^E  |passed|failed|^FAILED|error" | cut -c1-300 | head -40) > $O/pytest.log
(timeout 600 python bench.py 2>/dev/null | tail -1) > $O/bench.json
echo done > $O/finished
