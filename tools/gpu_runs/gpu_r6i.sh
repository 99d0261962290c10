#!/bin/bash
# round 6, call i: bench line with the space-backward roofline object, the re-bounded batch-32 parity test, whole GPU suite
set -u
O=gpurun_out/r6i
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity_bf16.py -q -x -s -k "batch_32" 2>&1 | grep -E "^\[bf16|^E  |passed|failed|^FAILED" | cut -c1-700 | head -20 > $O/tests_parity.txt
(timeout 400 python bench.py 2>$O/bench.err | grep '^{' | tail -1) > $O/bench.json
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | cut -c1-300 > $O/tests.txt
tail -c 1000 $O/bench.err > $O/bench.err.tail; rm $O/bench.err
echo done > $O/finished
