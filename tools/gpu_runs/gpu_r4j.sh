#!/bin/bash
# round 4, call j: LDS-DMA rings in the streaming kernels (fwd, dq, dkv) -- parity tests, A/B against register staging
# (variant bit 2) at the config-4 shape; the one-rank RCCL bench line (stdout kept whole this time)
set -u
O=gpurun_out/r4j
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_stream_attention.py tests/test_gpu_kernels.py -x -q > $O/pytest_stream.log 2>&1
echo "rc=$?" >> $O/pytest_stream.log
for v in 0 4; do
  for m in fwd bwd; do
    PROBE_STREAM_VARIANT=$v PROBE_F=16 PROBE_N=576 PROBE_H=16 timeout 120 python tools/probe_attn.py space $m 8 30 > $O/probe_${m}_v$v.log 2>&1
  done
done
LAVILA_BENCH_ONE_RANK_RCCL=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_one_rank_rccl.out 2> $O/bench_one_rank_rccl.err
echo "rc=$?" >> $O/bench_one_rank_rccl.err
timeout 600 python bench.py --model CLIP_OPENAI_TIMESFORMER_LARGE_336PX --frames 16 --batch 8 --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_config4.out 2> $O/bench_config4.err
echo "rc=$?" >> $O/bench_config4.err
echo done > $O/finished
