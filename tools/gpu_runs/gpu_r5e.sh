#!/bin/bash
# round 5, call e: same-box comparison of the round-4 tree (exported to _r4_tree/) with this tree, rider variants of the
# time backward inside the bench step, the graphed step under a process group (tests), GPU suite
set -u
O=gpurun_out/r5e
mkdir -p $O
export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_gpu_ddp.py -m gpu -q 2>&1 | grep -E "^E  |passed|failed|^FAILED|rror" | cut -c1-1500 | head -40) > $O/pytest_ddp.log
B="--steps 10 --warmup 3 --no-cpu-baseline"
export LAVILA_BENCH_GRAPH=0
run() { name=$1; shift; (env "$@" timeout 300 python bench.py $B 2>$O/ab_$name.err | grep '^{' | tail -1) > $O/ab_$name.json; }
runold() { name=$1; (cd _r4_tree && timeout 300 python bench.py $B 2>../$O/ab_$name.err | grep '^{' | tail -1) > $O/ab_$name.json; }
runold r4_tree_a
run r5_tree_a X=1
runold r4_tree_b
run r5_tree_b X=1
run r5_rider0 LAVILA_TIME_BWD_RIDER=0
run r5_rider1 LAVILA_TIME_BWD_RIDER=1
run r5_rider2 LAVILA_TIME_BWD_RIDER=2
run r5_tokens_off LAVILA_COLSUM_TOKENS=0
python - > $O/ab_summary.txt <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/r5e/ab_*.json')):
    try:
        d = json.load(open(f)); print(os.path.basename(f), d['ms_per_step'], d['value'], d['roofline']['frac'], d['config'].get('host_enqueue_ms_per_step'))
    except Exception as e:
        print(os.path.basename(f), 'FAILED', e)
PY
(timeout 1800 python -m pytest tests -m gpu -q 2>&1 | grep -E "^E  |passed|failed|^FAILED|rror" | cut -c1-600 | head -40) > $O/pytest_all.log
echo done > $O/finished
