"""Times the narrator (BASELINE configs[4]: VCLM_OPENAI_TIMESFORMER_BASE_GPT2, 4 x 224^2, one GPU) on synthetic clips with
random-init weights: encode_image, then generate() three ways --
  graph     key/value cache, one hipGraph replay per token          (the product path)
  eager     key/value cache, kernels launched one by one
  recompute the reference's schedule: the whole prefix through the decoder for every token (narrator.py:118-143)
and writes a summary (captions/s, ms per token) to stdout / --out. Sampling: top_k=1 unless --sample (then multinomial).
    python tools/probe_narrator.py --batch 64 --length 77 [--half] [--out profiles/r03_narrator_decode.txt]
"""
import argparse
import contextlib
import io
import json
import os
import sys
import time
import types
import warnings

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--length', type=int, default=77)
    ap.add_argument('--returns', type=int, default=1)
    ap.add_argument('--reps', type=int, default=3)
    ap.add_argument('--half', action='store_true', help='model.half() + images.half() (main_infer_narrator.py --use-half)')
    ap.add_argument('--sample', action='store_true')
    ap.add_argument('--skip-recompute', action='store_true')
    ap.add_argument('--modes', default='graph,eager,recompute', help='comma list of graph, eager, recompute')
    ap.add_argument('--out')
    a = ap.parse_args()
    from lavila.models import models
    with warnings.catch_warnings(), contextlib.redirect_stdout(io.StringIO()):
        warnings.simplefilter('ignore')
        torch.manual_seed(0)
        m = models.VCLM_OPENAI_TIMESFORMER_BASE_GPT2(gated_xattn=True, num_frames=4)
    with torch.no_grad():
        for b in m.text_decoder.transformer.h:                   # live gates (the shipped zeros switch the image path off)
            b.alpha_cattn.fill_(0.5)
            b.alpha_dense.fill_(0.5)
    m = m.cuda().eval()
    if a.half:
        m = m.half()
    tok = types.SimpleNamespace(bos_token_id=50256, eos_token_id=50256, pad_token_id=50256)
    clips = torch.randn(a.batch, 3, 4, 224, 224, device='cuda', dtype=torch.float16 if a.half else torch.float32)
    amp = contextlib.nullcontext() if a.half else torch.autocast('cuda', dtype=torch.bfloat16)
    kw = dict(max_text_length=a.length, num_return_sequences=a.returns)
    kw.update(dict(top_k=None, top_p=0.95, temperature=0.7) if a.sample else dict(top_k=1))   # main_infer_narrator.py:55-59
    res = {'batch': a.batch, 'length': a.length, 'returns': a.returns, 'dtype': 'fp16 in / bf16 compute' if a.half else 'f32 masters, bf16 autocast',
           'sampling': 'nucleus top_p=0.95, temperature 0.7' if a.sample else 'top_k=1', 'device': torch.cuda.get_device_name(0)}
    with torch.no_grad(), amp:
        t_enc, img = timed(lambda: m.encode_image(clips), a.reps)
        res['encode_image_ms'] = round(t_enc * 1e3, 2)
        outs = {}
        for name, g in (('graph', dict(kv_cache=True, graph=True)), ('eager', dict(kv_cache=True, graph=False)),
                        ('recompute', dict(kv_cache=False))):
            if (name == 'recompute' and a.skip_recompute) or name not in a.modes.split(','):
                continue
            torch.manual_seed(1)
            t, out = timed(lambda: m.generate(img, tok, **kw, **g), 1 if name == 'recompute' else a.reps)
            outs[name] = out
            steps = a.length - 1
            res[name] = {'generate_ms': round(t * 1e3, 1), 'ms_per_token_step': round(t * 1e3 / steps, 3),
                         'captions_per_s_decode_only': round(a.batch * a.returns / t, 1),
                         'captions_per_s_with_encoder': round(a.batch * a.returns / (t + t_enc), 1)}
        if not a.sample and 'graph' in outs:
            ref = outs['graph'][0]
            for name, (ids, ppl) in outs.items():
                same = (ids == ref).float().mean().item()
                res[name]['token_agreement_with_graph'] = round(same, 4)
        if 'recompute' in res and 'graph' in res:
            res['speedup_graph_vs_recompute'] = round(res['recompute']['generate_ms'] / res['graph']['generate_ms'], 2)
        if 'eager' in res and 'graph' in res:
            res['speedup_graph_vs_eager'] = round(res['eager']['generate_ms'] / res['graph']['generate_ms'], 2)
    # where the sampling kernel spends its time (phase clocks of lvl_sample_next_token's debug output), on logits
    # of the decoder's shape
    from lavila_amd.narrator import sample_next_token
    rows = a.batch * a.returns
    lg = (3 * torch.randn(rows, 50432, device='cuda')).bfloat16()[:, :50257]
    skw = dict(top_k=None, top_p=0.95, temperature=0.7) if a.sample else dict(top_k=1, top_p=None, temperature=1.0)
    for _ in range(2):
        out = sample_next_token(lg, debug=True, **skw)
    torch.cuda.synchronize()
    dbg = out[3].cpu()
    ph = dbg[:, 4:9].mean(0).tolist()
    res['sampler_phase_us'] = dict(zip(['load+max', 'stats', 'top_k', 'top_p', 'draw'], [round(v, 1) for v in ph]))
    res['sampler_workgroup_us'] = round(float(dbg[:, 4:9].sum(1).mean()), 1)
    start = dbg[:, 9]
    res['sampler_start_spread_us'] = round(float(start.max() - start.min()), 1)
    text = json.dumps(res, indent=1)
    print(text)
    if a.out:
        with open(a.out, 'w') as f:
            f.write(text + '\n')


if __name__ == '__main__':
    main()
