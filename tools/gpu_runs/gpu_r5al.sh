#!/bin/bash
# round 5, call al: final tree -- full GPU suite, smoke, default bench line
set -u
O=gpurun_out/r5al
mkdir -p $O
export TMPDIR=/tmp
timeout 430 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1
echo "pytest exit $?" >> $O/pytest_gpu.txt
timeout 100 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1
timeout 200 python bench.py > $O/bench.json 2> $O/bench.err
echo done > $O/finished
