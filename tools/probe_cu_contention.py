"""How does the step react when another kernel holds some compute units (what an RCCL collective's channel workgroups do
during the gradient all-reduce)? Runs bench.py's step loop while a background thread keeps `--spin-wgs` spin workgroups
resident on a second stream. usage: python tools/probe_cu_contention.py [spin_wgs] [bench args...]"""
import ctypes
import os
import subprocess
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
so = os.path.join(ROOT, 'tools', 'probes', 'libspin_probe.so')      # probe code stays out of the product lib dir
if '--build' in sys.argv:
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-fPIC', '-shared', '-w',
                           os.path.join(ROOT, 'tools', 'probes', 'spin.hip'), '-o', so])
    print('built', so)
    sys.exit(0)
import torch  # noqa: E402

wgs = int(sys.argv[1]) if len(sys.argv) > 1 else 16
lib = ctypes.CDLL(so)
lib.spin_launch.argtypes = [ctypes.c_int, ctypes.c_long, ctypes.c_void_p, ctypes.c_void_p]
stop = False


def spinner():
    torch.cuda.set_device(0)
    st = torch.cuda.Stream()
    sink = torch.zeros(1, dtype=torch.int32, device='cuda')
    while not stop:
        # ~3 ms per launch at ~2 GHz; a few in flight so that the units stay taken across kernel boundaries
        for _ in range(4):
            lib.spin_launch(wgs, 6_000_000, ctypes.c_void_p(sink.data_ptr()), ctypes.c_void_p(st.cuda_stream))
        st.synchronize()


import bench  # noqa: E402
sys.argv = ['bench.py', '--no-cpu-baseline', '--no-events'] + sys.argv[2:]
th = None
if wgs > 0:
    th = threading.Thread(target=spinner, daemon=True)
    th.start()
bench.main()
stop = True
