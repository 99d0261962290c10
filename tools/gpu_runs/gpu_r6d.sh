#!/bin/bash
# round 6, call d: activation checkpointing on the benched path (fixed: one ctx.saved_tensors access), configs[2] at local
# batch 256 x 16 frames with it, config 4 at larger local batches, GEMM store-policy A/B
set -u
O=gpurun_out/r6d
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_model.py -q -x -k "checkpoint" 2>&1 | grep -E "^E  |passed|failed|^FAILED" | cut -c1-600 | head -20 > $O/tests.txt
timeout 600 python tools/probe_gemm_store_policy.py 2>&1 | grep -v amdgpu.ids | cut -c1-400 > $O/store_policy.txt
timeout 600 python bench.py --frames 16 --batch 256 --checkpoint --steps 3 --warmup 1 --no-cpu-baseline > $O/config3_b256_16f_ckpt.json 2> $O/config3_b256_16f_ckpt.err
timeout 600 python bench.py --model CLIP_OPENAI_TIMESFORMER_LARGE_336PX --frames 16 --batch 16 --steps 3 --warmup 1 --no-cpu-baseline > $O/config4_b16.json 2> $O/config4_b16.err
timeout 600 python bench.py --model CLIP_OPENAI_TIMESFORMER_LARGE_336PX --frames 16 --batch 24 --steps 3 --warmup 1 --no-cpu-baseline > $O/config4_b24.json 2> $O/config4_b24.err
for f in $O/*.err; do tail -c 1500 $f > $f.tail; rm $f; done
echo done > $O/finished
