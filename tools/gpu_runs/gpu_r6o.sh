#!/bin/bash
# round 6, call o: VALU instruction rates (tools/probes/valu_rates.hip); serial kernel traces of the bench with the call-m library
# (only the GEMM file without SLP pairs) and the all-files -fno-slp-vectorize library: which kernels move which way
set -u
O=gpurun_out/r6o
mkdir -p $O
export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/probes/valu_rates.hip -o /tmp/valu_rates 2>/dev/null && timeout 300 /tmp/valu_rates > $O/valu_rates.txt 2>&1
L=lavila_amd/lib/liblavila_hip.so
cp $L /tmp/lavila_new.so
for v in base new; do
  if [ $v = base ]; then cp tools/probes/ab/liblavila_hip_base.so $L; else cp /tmp/lavila_new.so $L; fi
  cd /tmp
  LAVILA_TEXT_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-events > $GRAFT_REPO_ROOT/$O/prof_$v.log 2>&1
  cd $GRAFT_REPO_ROOT
  DB=$(find $O/prof -name "*.db" | head -1)
  [ -n "$DB" ] && python tools/kernel_stats.py $DB 7 > $O/kernel_stats_serial_$v.csv 2>$O/kernel_stats.err
  rm -rf $O/prof
done
cp /tmp/lavila_new.so $L
echo done > $O/finished
