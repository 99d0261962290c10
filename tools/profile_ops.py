"""torch.profiler view of bench steps in steady state: which ATen ops launch the small kernels (fills, casts, adds).
usage: python tools/profile_ops.py   (3 warm-up steps outside the profile, 2 profiled steps)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = ['bench.py', '--steps', '2', '--warmup', '3', '--no-cpu-baseline', '--no-events']
import bench  # noqa: E402  (enables the tuned GEMM table)
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

prof = profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=False)
orig_perf = bench.time.perf_counter
state = {'n': 0}


def hooked():            # bench calls perf_counter() right before and right after the timed steps
    state['n'] += 1
    if state['n'] == 1:
        prof.start()
    elif state['n'] == 2:
        torch.cuda.synchronize()
        prof.stop()
    return orig_perf()


bench.time.perf_counter = hooked
bench.main()
print(prof.key_averages().table(sort_by='self_cuda_time_total', row_limit=40, max_name_column_width=60))
print(prof.key_averages(group_by_input_shape=True).table(sort_by='count', row_limit=30, max_name_column_width=50))
