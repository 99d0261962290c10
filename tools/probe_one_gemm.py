"""Runs ONE GEMM shape a few times (for rocprofv3 PMC / trace passes): python tools/probe_one_gemm.py qkv own|lib [reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from lavila_amd import _cabi as C  # noqa: E402
from lavila_amd import ops  # noqa: E402

SHAPES = {'qkv': (2304, 768), 'proj': (768, 768), 'fc1': (3072, 768), 'fc2': (768, 3072), 'dqkv': (768, 2304)}
name, mode = sys.argv[1], sys.argv[2]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
N, K = SHAPES[name]
M = 256 * 785
x = torch.randn(M, K, device='cuda').bfloat16()
w = (torch.randn(N, K, device='cuda') * K ** -0.5).bfloat16()
b = torch.randn(N, device='cuda')
bb = b.bfloat16()
for _ in range(reps):
    y = ops.linear_tn_raw(x, w, b, C.EPI_BIAS) if mode == 'own' else F.linear(x, w, bb)
torch.cuda.synchronize()
print('done', y.float().abs().mean().item())
