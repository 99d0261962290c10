#!/bin/bash
# round 6, call n: -fno-slp-vectorize on every translation unit (v_pk_*_f32 is half rate): same-box bench A/B against the
# library of call m (GEMM file only), space / time backward kernel A/B, config-4 bench A/B, GPU suite
set -u
O=gpurun_out/r6n
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python tools/probe_space_bwd_ab.py space 2>&1 | grep -v amdgpu.ids > $O/attn_ab.txt
timeout 300 python tools/probe_space_bwd_ab.py time 2>&1 | grep -v amdgpu.ids >> $O/attn_ab.txt
tools/ab_library_swap.sh run $O/ab.txt --steps 10 --warmup 3
tools/ab_library_swap.sh run $O/ab_config4.txt --model CLIP_OPENAI_TIMESFORMER_LARGE_336PX --frames 16 --batch 8 --steps 4 --warmup 2
timeout 2400 python -m pytest tests -q -x -m gpu 2>&1 | grep -E "^E  |passed|failed|^FAILED" | cut -c1-600 | head -20 > $O/tests.txt
echo done > $O/finished
