#!/bin/bash
# round 5, call a: validation of the parity work (format-2 "spread" fixtures incl. the true 16-frame shapes, the narrator's
# float32 decoder on the own f32-class kernels under forbid_library_gemm, eval between graph replays, fp8 end-to-end bound),
# the default bench of the round's first tree, and the bisect of the one-rank RCCL / DDP overhead
set -u
O=gpurun_out/r5a
mkdir -p $O
export TMPDIR=/tmp
(timeout 300 python __graft_entry__.py smoke 2>&1 | tail -5) > $O/smoke.log

(timeout 1800 python -m pytest tests -m gpu -q -s 2>&1 | grep -E "^E  |passed|failed|^FAILED|rror|f32-class|fp8 end" | cut -c1-700 | head -100) > $O/pytest_all.log
(timeout 600 python bench.py 2>$O/bench.err | grep '^{' | tail -1) > $O/bench.json
B="--steps 10 --warmup 3 --no-cpu-baseline"
export LAVILA_BENCH_GRAPH=0
run() { name=$1; shift; (env "$@" timeout 300 python bench.py $B 2>$O/bisect_$name.err | grep '^{' | tail -1) > $O/bisect_$name.json; }
run plain X=1
run plain_serial LAVILA_TEXT_STREAM=0
run group_only LAVILA_BENCH_ONE_RANK_RCCL=1 LAVILA_BENCH_NO_DDP=1 LAVILA_DYNAMIC_TILES=0
run group_ddp_static LAVILA_BENCH_ONE_RANK_RCCL=1 LAVILA_DYNAMIC_TILES=0
run group_ddp_dynamic LAVILA_BENCH_ONE_RANK_RCCL=1
run group_ddp_static_serial LAVILA_BENCH_ONE_RANK_RCCL=1 LAVILA_DYNAMIC_TILES=0 LAVILA_TEXT_STREAM=0
run plain_again X=1
python - > $O/bisect_summary.txt <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/r5a/bisect_*.json')):
    try:
        d = json.load(open(f)); print(os.path.basename(f), d['ms_per_step'], d['value'], d['config'].get('host_enqueue_ms_per_step'))
    except Exception as e:
        print(os.path.basename(f), 'FAILED', e)
PY
echo done > $O/finished
