"""Shared test helpers: build the lavila_amd model for a golden config, oracle-side slab maths."""
import os
import sys

import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import oracle as O  # noqa: E402


def build_model(c, quiet=True):
    """Same architecture as oracle/gen_golden.build_reference_model, built from lavila_amd classes through
    the reference import paths."""
    import contextlib
    import io
    from lavila.models import models
    from lavila.models.openai_model import QuickGELU
    from lavila.models.timesformer import SpaceTimeTransformer
    with contextlib.redirect_stdout(io.StringIO()) if quiet else contextlib.nullcontext():
        vis = SpaceTimeTransformer(
            img_size=c['img'], patch_size=c['patch'], embed_dim=c['dim'], depth=c['depth'], num_heads=c['heads'],
            num_frames=c['frames'], time_init='zeros', attention_style='frozen-in-time', ln_pre=True,
            act_layer=QuickGELU, is_tanh_gating=c['gated'])
        vis.head = nn.Identity()
        vis.pre_logits = nn.Identity()
        vis.fc = nn.Identity()
        model = models.CLIP(
            embed_dim=c['embed'], vision_width=c['dim'], vision_model=vis, context_length=77,
            vocab_size=c['vocab'], transformer_width=c['t_width'], transformer_heads=c['t_heads'],
            transformer_layers=c['t_layers'], tempearture_init=0.07)
    return model


def fixture_weights(fx):
    """The procedural weights a model fixture was generated with (format-2 / "spread" fixtures: oracle.procedural_weights
    with spread=True)."""
    return O.procedural_weights(fx['shapes'], seed=fx['weight_seed'], spread=fx['config'].get('spread', False))


def check_fixture_gradients(fx, grads, rtol, norm_rtol, tag=''):
    """Parameter gradients against a model fixture. Format 1 (rounds 1-4): ten full tensors at atol 1e-4 + every norm.
    Format 2 (round 5): every stored full gradient (1-D tensors of six blocks, the embeddings, the LayerNorms: 69 tensors)
    and four rows of every stored weight gradient, each on ITS OWN scale -- max |got - ref| <= rtol * max |ref| and relative
    L2 <= rtol -- so that a gradient of magnitude 1e-4 is checked as tightly as one of magnitude 1, and a direction error
    cannot hide behind a matching norm; plus every parameter's norm. Returns the worst relative distance seen."""
    worst = 0.0
    if fx.get('format', 1) == 2:
        items = [(k, grads[k].detach().float().cpu(), g) for k, g in fx['grads'].items()]
        for k, (rows, g) in fx['grad_slices'].items():
            got = grads[k].detach().float().cpu()
            got = got.reshape(-1, got.shape[-1]) if k in ('visual.pos_embed', 'positional_embedding') else got.reshape(got.shape[0], -1)
            items.append((k + '[rows]', got[rows], g))
        assert len(items) >= 40
        bad = []
        for k, got, ref in items:
            scale = ref.abs().max().item()
            assert scale > 0, f'{tag}{k}: the reference gradient is identically zero -- nothing is checked'
            d = (got - ref).abs().max().item() / scale
            l2 = ((got - ref).norm() / ref.norm()).item()
            worst = max(worst, d, l2)
            if not (d <= rtol and l2 <= rtol):
                bad.append((max(d, l2), f'{k}: max |d| / max |ref| = {d:.2e}, relative L2 = {l2:.2e}'))
        assert not bad, (f'{tag}{len(bad)} of {len(items)} gradient tensors beyond {rtol:.0e} on their own scale; worst: ' +
                         '; '.join(m for _, m in sorted(bad, reverse=True)[:6]))
    else:
        for k, gref in fx['grads'].items():
            torch.testing.assert_close(grads[k].detach().float().cpu(), gref, atol=1e-4, rtol=rtol, msg=lambda m: f'{tag}{k}: {m}')
    for k, n in fx.get('grad_norms', {}).items():
        got = grads[k].detach().float().norm().item()
        assert abs(got - n) <= norm_rtol * n + 1e-6, (tag, k, got, n)
    return worst


def oracle_slab_forward(img_all, txt_all, scale, B, row0):
    """CPU restatement of lvl_clip_loss_fwd (stats [2,B,4], argmax [2,B]) from oracle.clip_logits."""
    li = O.clip_logits(img_all.float(), txt_all.float(), scale.float())          # [G,G] logits_per_image
    slabs = torch.stack([li[row0:row0 + B], li.t()[row0:row0 + B]])             # [2,B,G]
    lse = torch.logsumexp(slabs, -1)
    idx = torch.arange(row0, row0 + B)
    diag = slabs[:, torch.arange(B), idx]
    p = torch.softmax(slabs, -1)
    expect = (p * slabs).sum(-1)
    stats = torch.stack([lse, diag, expect, slabs.max(-1).values], -1)
    return stats, slabs.argmax(-1).to(torch.int32)


def oracle_slab_backward(img_all, txt_all, lse_all, scale, upstream, coef, B, row0, rows_only=False):
    """CPU restatement of lvl_clip_loss_bwd via autograd on the oracle's full loss:
    coef*upstream*d(sum of both CE sums)/d(local rows) = coef*upstream*2G * d(loss)/d(local rows).
    rows_only: the local rows' own two cross-entropy sums against CONSTANT gathered partners (loss.py:34-43,86-88)."""
    G = img_all.shape[0]
    if rows_only:
        with torch.enable_grad():
            il = img_all[row0:row0 + B].float().clone().requires_grad_(True)
            tl = txt_all[row0:row0 + B].float().clone().requires_grad_(True)
            s = scale.float().reshape(())
            labels = torch.arange(row0, row0 + B)
            tot = (torch.nn.functional.cross_entropy(s * il @ txt_all.float().t(), labels, reduction='sum') +
                   torch.nn.functional.cross_entropy(s * tl @ img_all.float().t(), labels, reduction='sum'))
            gi, gt = torch.autograd.grad(tot, [il, tl])
        k = coef * upstream.reshape(())
        return (k * gi).contiguous(), (k * gt).contiguous()
    with torch.enable_grad():          # may be called from inside an autograd backward (grad mode off)
        ia = img_all.float().clone().requires_grad_(True)
        ta = txt_all.float().clone().requires_grad_(True)
        loss = O.clip_loss(ia, ta, scale.float().reshape(()))['loss']
        gi, gt = torch.autograd.grad(loss, [ia, ta])
    k = coef * upstream.reshape(()) * 2 * G
    return (k * gi[row0:row0 + B]).contiguous(), (k * gt[row0:row0 + B]).contiguous()


def oracle_ssl_slab_forward(img_all, txt_all, ind_all, scales3, B, row0):
    """CPU restatement of lvl_ssl_clip_loss_fwd: stats [2,B,8] = {lse, diag logit, E0, E1, E2, diag dot, max, 0}
    with E_k = sum_j softmax_ij * dot_ij * [ind_i + ind_j == k]; argmax [2,B]."""
    ind = ind_all.long()
    bucket = ind[:, None] + ind[None, :]
    dots = img_all.float() @ txt_all.float().t()
    li = scales3.float()[bucket] * dots
    idx = torch.arange(row0, row0 + B)
    rows = torch.arange(B)
    out = []
    for L, D, Bk in ((li[row0:row0 + B], dots[row0:row0 + B], bucket[row0:row0 + B]),
                     (li.t()[row0:row0 + B], dots.t()[row0:row0 + B], bucket.t()[row0:row0 + B])):
        p = torch.softmax(L, -1)
        ek = [(p * D * (Bk == k)).sum(-1) for k in range(3)]
        out.append(torch.stack([torch.logsumexp(L, -1), L[rows, idx], ek[0], ek[1], ek[2], D[rows, idx],
                                L.max(-1).values, torch.zeros(B)], -1))
    slabs = torch.stack([li[row0:row0 + B], li.t()[row0:row0 + B]])
    return torch.stack(out), slabs.argmax(-1).to(torch.int32)


def oracle_ssl_slab_backward(img_all, txt_all, ind_all, lse_all, scales3, upstream, coef, B, row0):
    """CPU restatement of lvl_ssl_clip_loss_bwd via autograd on oracle.ssl_clip_loss (scales3 = {pseudo, geo, real})."""
    G = img_all.shape[0]
    with torch.enable_grad():
        ia = img_all.float().clone().requires_grad_(True)
        ta = txt_all.float().clone().requires_grad_(True)
        loss = O.ssl_clip_loss(ia, ta, ind_all.long(), scales3[2].float(), scales3[0].float())['loss']
        gi, gt = torch.autograd.grad(loss, [ia, ta])
    k = coef * upstream.reshape(()) * 2 * G
    return (k * gi[row0:row0 + B]).contiguous(), (k * gt[row0:row0 + B]).contiguous()
