#!/bin/bash
set -u
O=gpurun_out/r3h
mkdir -p $O
export TMPDIR=/tmp
(timeout 1200 python -m pytest tests -m gpu -q 2>&1 | grep -E "^E  |passed|failed|^FAILED" | cut -c1-300 | head -30) > $O/pytest.log
run() { tag=$1; shift
  env "$@" timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_$tag.json
  python -c "import json,sys; d=json.load(open('$O/bench_$tag.json')); print('$tag', d['value'], d['ms_per_step'], d['config']['final_loss'], d['roofline']['achieved'], d['roofline_wgrad']['achieved'])" >> $O/ab.txt
}
run cls1 LAVILA_CLS_LAST=1
run cls0 LAVILA_CLS_LAST=0
run cls1b LAVILA_CLS_LAST=1
run cls0b LAVILA_CLS_LAST=0
echo done > $O/finished
