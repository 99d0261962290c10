"""How fast is the CPU oracle (oracle/oracle.py, what bench.py's cpu_baseline leg times on the GPU box: kind "port") next to
the REFERENCE'S OWN modules on the same host, same threads, same batch? The reference is only present in the build
container (/root/reference), so this runs there, once, and the ratio is recorded under profiles/ (VERDICT r3 item 9).
usage: python tools/cpu_baseline_ratio.py [batch] [threads]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from oracle import oracle as O  # noqa: E402
from oracle.gen_golden import CONFIGS, build_reference_model, synthetic_inputs  # noqa: E402
from oracle.ref_import import load_reference  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4
threads = int(sys.argv[2]) if len(sys.argv) > 2 else min(32, os.cpu_count() or 1)
torch.set_num_threads(threads)
c = dict(CONFIGS['config2_tsfb_224_b8_spread'], batch=batch)
ref = load_reference()
torch.manual_seed(0)
model = build_reference_model(ref, c)
shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
weights = O.procedural_weights(shapes, seed=7)
model.load_state_dict(weights, strict=True)
model.train()
video, tokens = synthetic_inputs(c)
crit = ref.loss.CLIPLoss(use_vissl=False, cache_labels=True, rank=0, world_size=1)


def run_reference():
    out = model(video, tokens, norm_embed=True)
    crit(out)['loss'].backward()
    model.zero_grad(set_to_none=True)


w = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in weights.items()}


def run_port():
    out = O.clip_forward(video, tokens, w, c['heads'], c['t_heads'], norm_embed=True)
    O.clip_loss(out['image_embed'], out['text_embed'], out['logit_scale'])['loss'].backward()
    for v in w.values():
        v.grad = None


def timed(fn, n=3):
    fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return sum(ts) / len(ts), min(ts)


r_mean, r_best = timed(run_reference)
p_mean, p_best = timed(run_port)
cpu = 'unknown'
for line in open('/proc/cpuinfo'):
    if line.startswith('model name'):
        cpu = line.split(':', 1)[1].strip()
        break
res = {'what': 'fwd + loss + bwd of TSF-B/16 4x224^2 + text tower (config 2 shapes), float32, batch %d, %d torch threads, '
               '1 warm-up + 3 timed iterations each, same process' % (batch, threads),
       'cpu': f'{cpu} ({os.cpu_count()} logical cores)',
       'reference_modules': {'pairs_per_s': round(batch / r_mean, 4), 'mean_s': round(r_mean, 3), 'best_s': round(r_best, 3)},
       'oracle_port': {'pairs_per_s': round(batch / p_mean, 4), 'mean_s': round(p_mean, 3), 'best_s': round(p_best, 3)},
       'port_over_reference': round(r_mean / p_mean, 3)}
print(json.dumps(res, indent=1))
