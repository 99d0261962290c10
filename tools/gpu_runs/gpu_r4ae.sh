#!/bin/bash
# round 4, call ae: HBM ceiling for the read/write mixes of the row kernels (tools/probes/stream_mix.hip)
set -u
O=gpurun_out/r4ae
mkdir -p $O
timeout 120 tools/probes/stream_mix > $O/stream_mix.txt 2>&1
echo "rc=$?" >> $O/stream_mix.txt
echo done > $O/finished
