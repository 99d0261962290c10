#!/bin/bash
# round 4, call z: residual epilogues -- the tower test against a float32 reference, the full GPU suite with the flag on
set -u
O=gpurun_out/r4z
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "residual" > $O/pytest_residual.log 2>&1
echo "rc=$?" >> $O/pytest_residual.log
(LAVILA_RESIDUAL_EPILOGUE=1 timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -E "^E  |passed|failed|^FAILED|error" | cut -c1-300 | head -40) > $O/pytest_full_flag_on.log
echo done > $O/finished
