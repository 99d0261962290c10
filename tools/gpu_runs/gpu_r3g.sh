#!/bin/bash
set -u
O=gpurun_out/r3g
mkdir -p $O
export TMPDIR=/tmp
(timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -8) > $O/pytest.log
run() { tag=$1; shift
  env "$@" timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_$tag.json
  python -c "import json,sys; d=json.load(open('$O/bench_$tag.json')); print('$tag', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline_wgrad']['achieved'])" >> $O/ab.txt
}
run prio_low LAVILA_TEXT_STREAM_PRIORITY=low
run prio_normal LAVILA_TEXT_STREAM_PRIORITY=normal
run prio_low2 LAVILA_TEXT_STREAM_PRIORITY=low
run prio_normal2 LAVILA_TEXT_STREAM_PRIORITY=normal
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range())" >> $O/ab.txt 2>&1
echo done > $O/finished
