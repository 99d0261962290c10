#!/bin/bash
# round 4, call ad: narrator bench lines on the final tree (the encoder now runs the residual epilogues; decode session
# handling changed this round), beam-search timing for the record
set -u
O=gpurun_out/r4ad
mkdir -p $O
export TMPDIR=/tmp
(timeout 600 python bench.py --workload narrator --steps 4 --warmup 1 2>$O/narr_r10.err | grep '^{' | tail -1) > $O/bench_narrator_n10.json
(timeout 600 python bench.py --workload narrator --returns 1 --steps 4 --warmup 1 2>$O/narr_r1.err | grep '^{' | tail -1) > $O/bench_narrator_n1.json
echo done > $O/finished
