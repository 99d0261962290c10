#!/bin/bash
# round 6, call r: plain LayerNorm forward at 56 VGPRs (8 waves/SIMD): grid sweep with tools/probe_rowops.py
set -u
O=gpurun_out/r6r
mkdir -p $O
export TMPDIR=/tmp
for b in 0 2048 4096 6144 8192; do
  echo "== LAVILA_LN_FWD_BLOCKS=$b" >> $O/rowops.txt
  LAVILA_LN_FWD_BLOCKS=$b timeout 300 python tools/probe_rowops.py 256 20 2>&1 | grep -E "ln_fwd" >> $O/rowops.txt
done
cp tools/probes/ab/liblavila_hip_base.so /tmp/base.so
echo "== base library" >> $O/rowops.txt
cp lavila_amd/lib/liblavila_hip.so /tmp/new.so; cp /tmp/base.so lavila_amd/lib/liblavila_hip.so
timeout 300 python tools/probe_rowops.py 256 20 2>&1 | grep -E "ln_fwd" >> $O/rowops.txt
cp /tmp/new.so lavila_amd/lib/liblavila_hip.so
echo done > $O/finished
