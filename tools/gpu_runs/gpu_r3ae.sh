#!/bin/bash
set -u
O=gpurun_out/r3ae
mkdir -p $O
export TMPDIR=/tmp
(timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3) > $O/smoke.log
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -E "^E  |passed|failed|^FAILED|error" | cut -c1-300 | head -40) > $O/pytest.log
(timeout 600 python bench.py 2>/dev/null | tail -1) > $O/bench.json
echo done > $O/finished
