#!/bin/bash
# round 4, call ag: flakiness check -- the GPU suite three times in a row on the final tree, smoke once
set -u
O=gpurun_out/r4ag
mkdir -p $O
export TMPDIR=/tmp
for i in 1 2 3; do
  (timeout 900 python -m pytest tests -m gpu -q 2>&1 | grep -E "^E  |passed|failed|^FAILED|error" | cut -c1-300 | head -20) > $O/pytest_$i.log
done
(timeout 300 python __graft_entry__.py smoke 2>&1 | tail -4) > $O/smoke.log
echo done > $O/finished
