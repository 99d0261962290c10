#!/bin/bash
# round 6, call z: time-attention forward with every row of a location requested up front and no conditional store in the loop:
# attention tests, kernel trace A/B (serial), bench A/B
set -u
O=gpurun_out/r6z
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity_bf16.py -q -x -k "attention or attn or divided" 2>&1 | grep -E "^E  |passed|failed|^FAILED" | cut -c1-400 | head > $O/tests.txt
tools/ab_library_swap.sh run $O/ab.txt --steps 10 --warmup 3
L=lavila_amd/lib/liblavila_hip.so
cp $L /tmp/lavila_new.so
for v in base new; do
  if [ $v = base ]; then cp tools/probes/ab/liblavila_hip_base.so $L; else cp /tmp/lavila_new.so $L; fi
  cd /tmp
  LAVILA_TEXT_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-events > $GRAFT_REPO_ROOT/$O/prof_$v.log 2>&1
  cd $GRAFT_REPO_ROOT
  DB=$(find $O/prof -name "*.db" | head -1)
  [ -n "$DB" ] && python tools/kernel_stats.py $DB 7 2>/dev/null | grep -E "time_fwd|time_bwd|ln_fwd|ln_bwd|total kernel" > $O/kernel_stats_$v.txt
  rm -rf $O/prof
done
cp /tmp/lavila_new.so $L
echo done > $O/finished
