#!/bin/bash
# round 4, call y: LVL_EPI_BIAS_RESIDUAL (projection / fc2 + residual in the GEMM epilogue): kernel tests, block and tower
# equivalence, same-box A/B of the default bench with the integration off / on
set -u
O=gpurun_out/r4y
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity_bf16.py tests/test_gpu_f32_class.py tests/test_gpu_kernels.py -x -q -k "linear or residual or gemm or mlp" > $O/pytest.log 2>&1
echo "rc=$?" >> $O/pytest.log
for v in 0 1 0 1; do
  echo "residual_epilogue=$v $(LAVILA_RESIDUAL_EPILOGUE=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-events 2>/dev/null | grep '^{' | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["config"]["final_loss"])')" >> $O/bench_ab.txt
done
LAVILA_RESIDUAL_EPILOGUE=1 timeout 1200 python -m pytest tests -m gpu -q -x > $O/pytest_full_flag_on.log 2>&1
echo "rc=$?" >> $O/pytest_full_flag_on.log
echo done > $O/finished
