#!/bin/bash
# round 6, call c: BASELINE configs[2] at its real per-GPU shape (TSF-B 16 x 224^2, local batch 256) with peak memory --
# without and with activation checkpointing --, the half-batch shape, and this round's starting records of config 4 and
# the narrator
set -u
O=gpurun_out/r6c
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python bench.py --frames 16 --batch 256 --steps 3 --warmup 1 --no-cpu-baseline > $O/config3_b256_16f_plain.json 2> $O/config3_b256_16f_plain.err
timeout 600 python bench.py --frames 16 --batch 256 --checkpoint --steps 3 --warmup 1 --no-cpu-baseline > $O/config3_b256_16f_ckpt.json 2> $O/config3_b256_16f_ckpt.err
timeout 600 python bench.py --frames 16 --batch 128 --steps 3 --warmup 1 --no-cpu-baseline > $O/config3_b128_16f_plain.json 2> $O/config3_b128_16f_plain.err
timeout 600 python bench.py --frames 16 --batch 64 --steps 4 --warmup 1 --no-cpu-baseline > $O/config3_b64_16f_plain.json 2> $O/config3_b64_16f_plain.err
timeout 600 python bench.py --model CLIP_OPENAI_TIMESFORMER_LARGE_336PX --frames 16 --batch 8 --steps 4 --warmup 1 --no-cpu-baseline > $O/config4.json 2> $O/config4.err
timeout 600 python bench.py --workload narrator --returns 1 --steps 4 --warmup 1 --no-cpu-baseline > $O/narrator_n1.json 2> $O/narrator_n1.err
timeout 600 python bench.py --workload narrator --returns 10 --steps 4 --warmup 1 --no-cpu-baseline > $O/narrator_n10.json 2> $O/narrator_n10.err
for f in $O/*.err; do tail -c 1500 $f > $f.tail; rm $f; done
echo done > $O/finished
