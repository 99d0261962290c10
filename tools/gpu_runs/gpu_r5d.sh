#!/bin/bash
# round 5, call d: the segmented graphed step under a process group (2 ranks gloo / 1 rank RCCL), default bench with the
# measured rider default, kernel traces of the plain and of the one-rank-DDP step (what DDP's +1.9 % consists of)
set -u
O=gpurun_out/r5d
mkdir -p $O
export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_gpu_ddp.py tests/test_gpu_graph_step.py tests/test_gpu_kernels.py -m gpu -q -x 2>&1 | grep -E "^E  |passed|failed|^FAILED|rror" | cut -c1-1500 | head -60) > $O/pytest.log
(timeout 600 python bench.py --steps 20 --warmup 5 2>$O/bench.err | grep '^{' | tail -1) > $O/bench.json
export LAVILA_BENCH_GRAPH=0
for mode in plain ddp; do
  cd /tmp
  export LAVILA_TEXT_STREAM=0
  if [ $mode = ddp ]; then export LAVILA_BENCH_ONE_RANK_RCCL=1 LAVILA_DYNAMIC_TILES=0; else unset LAVILA_BENCH_ONE_RANK_RCCL LAVILA_DYNAMIC_TILES; fi
  timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_$mode -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-events > $GRAFT_REPO_ROOT/$O/prof_$mode.log 2>&1
  cd $GRAFT_REPO_ROOT
  DB=$(find $O/prof_$mode -name "*.db" | head -1)
  [ -n "$DB" ] && python tools/kernel_stats.py $DB 7 > $O/bench_kernel_stats_${mode}_serial.csv 2>$O/kernel_stats_$mode.err
  rm -rf $O/prof_$mode
done
unset LAVILA_TEXT_STREAM LAVILA_BENCH_ONE_RANK_RCCL LAVILA_DYNAMIC_TILES
echo done > $O/finished
