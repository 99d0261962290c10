#!/bin/bash
set -u
O=gpurun_out/r3v
mkdir -p $O
export TMPDIR=/tmp
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -E "^E  |passed|failed|^FAILED|error" | cut -c1-300 | head -40) > $O/pytest.log
(timeout 600 python bench.py 2>/dev/null | tail -1) > $O/bench.json
echo done > $O/finished
