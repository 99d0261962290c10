#!/bin/bash
# round 6, call h: embedding backward with the 8-range duplicate sum (tests + timing), batch-32 parity test with its numbers
set -u
O=gpurun_out/r6h
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -x -k "text_embedding" 2>&1 | grep -E "^E  |passed|failed|^FAILED" | cut -c1-400 | head > $O/tests_embed.txt
timeout 300 python tools/probe_text_embed.py 2>&1 | grep -v amdgpu.ids > $O/text_embed_timing.txt
timeout 600 python -m pytest tests/test_gpu_parity_bf16.py tests/test_gpu_graph_step.py -q -x -s -k "batch_32 or free_device_memory" 2>&1 | grep -E "^\[bf16|^E  |passed|failed|^FAILED" | cut -c1-700 | head -20 > $O/tests_parity.txt
echo done > $O/finished
