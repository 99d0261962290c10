#!/bin/bash
# round 6, call y: LayerNorm kernels with explicit fused operations (general == exact to the bit): tests, bench A/B
set -u
O=gpurun_out/r6y
mkdir -p $O
export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -q 2>&1 | grep -E "^E  |passed|failed|^FAILED" | cut -c1-400 | head > $O/tests.txt
python - > $O/bitexact.txt 2>&1 <<'PY'
import os, subprocess, sys
code = r'''
import torch, sys
sys.path.insert(0, '.')
from lavila_amd import ops
torch.manual_seed(0)
x = torch.randn(5000, 768, device='cuda').bfloat16(); y = torch.randn_like(x); g = torch.randn(768, device='cuda'); b = torch.randn(768, device='cuda'); yb = torch.randn(768, device='cuda')
dy = torch.randn_like(x); dadd = torch.randn_like(x)
outs = []
h, _, mean, rstd = ops.layernorm_fwd_raw(x, None, None, g, b, 1e-5, False); outs += [h, mean, rstd]
h2, _, m2, r2 = ops.layernorm_fwd_raw(x, y, yb, g, b, 1e-5, False); outs += [h2, m2, r2]
outs += list(t for t in ops.layernorm_bwd_raw(dy, x, None, None, g, mean, rstd, dadd, True) if t is not None)
outs += list(t for t in ops.layernorm_bwd_raw(dy, x, y, yb, g, m2, r2, None, True) if t is not None)
outs += list(t for t in ops.layernorm_bwd_raw(dy, x, y, yb, g, m2, r2, dadd, True, True) if t is not None)
torch.save([o.cpu() for o in outs], sys.argv[1])
'''
for e in ('0', '1'):
    subprocess.run([sys.executable, '-c', code, f'/tmp/ln_{e}.pt'], env=dict(os.environ, LAVILA_LN_EXACT=e), check=True)
import torch
a, b = torch.load('/tmp/ln_0.pt'), torch.load('/tmp/ln_1.pt')
print('tensors', len(a), 'all bit-equal general vs exact:', all(torch.equal(p, q) for p, q in zip(a, b)))
for i, (p, q) in enumerate(zip(a, b)):
    if not torch.equal(p, q): print(i, (p.float() - q.float()).abs().max().item())
PY
for e in 0 1 0 1; do
  LAVILA_LN_EXACT=$e timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-events 2>/dev/null | grep '^{' | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("LN_EXACT='$e'", d["value"], d["ms_per_step"])' >> $O/ab.txt
done
echo done > $O/finished
