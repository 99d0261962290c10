#!/bin/bash
set -u
O=gpurun_out/r3aa
mkdir -p $O
export TMPDIR=/tmp
run() { tag=$1; shift
  env "$@" timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_$tag.json
  python -c "import json,sys; d=json.load(open('$O/bench_$tag.json')); print('$tag', d['value'], d['ms_per_step'], d['config']['final_loss'])" >> $O/ab.txt
}
run base LAVILA_WGRAD_STREAM=0
run side LAVILA_WGRAD_STREAM=1
run side_dyn LAVILA_WGRAD_STREAM=1 LAVILA_DYNAMIC_TILES=1
run base2 LAVILA_WGRAD_STREAM=0
run side2 LAVILA_WGRAD_STREAM=1
(LAVILA_WGRAD_STREAM=1 timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_parity_bf16.py -m gpu -q -x 2>&1 | tail -5 | cut -c1-300) > $O/pytest_side.log
echo done > $O/finished
