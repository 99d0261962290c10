#!/bin/bash
# round 4, call h: final validation -- smoke, full GPU suite (incl. the one-rank RCCL test), default bench, the bench on a
# one-rank RCCL group (DDP + collectives on the real library)
set -u
O=gpurun_out/r4h
mkdir -p $O
export TMPDIR=/tmp
(timeout 300 python __graft_entry__.py smoke 2>&1 | tail -4) > $O/smoke.log
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -E "^E  |passed|failed|^FAILED|error" | cut -c1-400 | head -40) > $O/pytest.log
(timeout 600 python bench.py 2>/dev/null | tail -1) > $O/bench.json
(LAVILA_BENCH_ONE_RANK_RCCL=1 timeout 600 python bench.py --no-cpu-baseline 2>$O/bench_rccl.err | tail -1) > $O/bench_one_rank_rccl.json
echo done > $O/finished
