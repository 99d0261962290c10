"""torch.profiler view of one bench step: which ATen ops launch the small kernels (fills, casts, adds)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = ['bench.py', '--steps', '1', '--warmup', '1', '--no-cpu-baseline', '--no-events']
import bench  # noqa: E402  (enables the tuned GEMM table)
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

orig_main = bench.main
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    orig_main()
print(prof.key_averages().table(sort_by='self_cuda_time_total', row_limit=45, max_name_column_width=60))
print(prof.key_averages(group_by_input_shape=True).table(sort_by='count', row_limit=25, max_name_column_width=50))
