"""GPU (-m gpu): the drop-in boundary as the reference's OTHER callers use it:
eval_zeroshot.py --use-half (model.half() + images.half(), no autocast), the narrator-style [B,F,C,H,W] entry of
forward_features, the gated / stochastic-depth block variants, out-of-band parameter writes and hipGraph replay
after a weight update."""
import contextlib
import io
import warnings

import pytest
import torch

from conftest import load_golden
from helpers import build_model
from oracle import oracle as O
from oracle.gen_golden import synthetic_inputs

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _rel(a, b):
    return ((a.float().cpu() - b.float()).norm() / b.float().norm().clamp_min(1e-30)).item()


def _tiny():
    fx = load_golden('model_tiny_p16.pt')
    c = fx['config']
    model = build_model(c)
    model.load_state_dict(O.procedural_weights(fx['shapes'], seed=fx['weight_seed']), strict=True)
    return fx, c, model


def test_use_half_eval_recipe_tiny():
    """Exactly what validate_zeroshot / get_similarity_matrix / validate_mcq do under --use-half
    (eval_zeroshot.py:210-261, 291-334, 337-354; docs/PRETRAIN.md:89): model.eval(); model = model.half();
    encode_text(tokens); encode_image(images.half()); x / x.norm(); .cpu().numpy(). fp16 parameters and clips are
    computed in bf16 (the kernels' half type) and fp16 comes back. Bound as for the bf16 autocast test of the same
    model (2 blocks: eps*sqrt(10*2) ~ 5e-3 relative; 4e-2 absolute used there) -- fp16 weight rounding (2^-11) is below
    the bf16 rounding (2^-9) of the very next operand load."""
    fx, c, model = _tiny()
    model.to(DEV)
    video, tokens = synthetic_inputs(c, seed=fx['input_seed'])
    model.eval()
    model = model.half()                                        # eval_zeroshot.py:212-213
    assert model.visual.blocks[0].attn.qkv.weight.dtype == torch.float16
    with torch.no_grad():
        te = model.encode_text(tokens.to(DEV))                  # :237-239
        images = video.to(DEV).half()                           # :256-257
        ie = model.encode_image(images)                         # :261
        feat = model.encode_image(images, apply_project=False)
    assert te.dtype == torch.float16 and ie.dtype == torch.float16 and feat.dtype == torch.float16
    ie = ie / ie.norm(dim=-1, keepdim=True)
    te = te / te.norm(dim=-1, keepdim=True)
    sim = ie.cpu().numpy() @ te.cpu().numpy().T                # :318-331
    assert sim.shape == (c['batch'], c['batch'])
    torch.testing.assert_close(ie.float().cpu(), fx['image_embed'], atol=4e-2, rtol=4e-2)
    torch.testing.assert_close(te.float().cpu(), fx['text_embed'], atol=4e-2, rtol=4e-2)
    # the same model used as a plain module (classifier-style call of the tower, models.py:40)
    with torch.no_grad():
        f2 = model.visual(images)
        allt = model.visual.forward_features(images.permute(0, 2, 1, 3, 4).contiguous(), cls_at_last=False)
    assert f2.dtype == torch.float16 and torch.equal(f2, feat)
    assert allt.dtype == torch.float16 and allt.shape == fx['features_all_tokens'].shape
    torch.testing.assert_close(allt.float().cpu(), fx['features_all_tokens'], atol=4e-2, rtol=4e-2)
    # state_dict stays a faithful fp16 image of the weights (load_state_dict(strict=True) round trip, eval_zeroshot.py:97)
    sd = model.state_dict()
    assert all(v.dtype == torch.float16 for v in sd.values() if v.is_floating_point())
    # a float32 model fed fp16 clips (frames.half() before the model was converted) still answers, in float32
    fx2, c2, m32 = _tiny()
    m32.to(DEV).eval()
    with torch.no_grad():
        ie32 = m32.encode_image(images)
    assert ie32.dtype == torch.float16 or ie32.dtype == torch.float32
    torch.testing.assert_close(O.l2_normalize(ie32.float().cpu()), fx['image_embed'], atol=4e-2, rtol=4e-2)


def test_use_half_eval_recipe_tsfb():
    """The same recipe on CLIP_OPENAI_TIMESFORMER_BASE at 4 x 224^2 (every kernel of the benched configuration: MFMA
    GEMMs on bf16 copies of the fp16 weights, 197-key space attention, register time attention, trimmed text tower)
    against the float32 oracle on the float32 weights. Bound: the depth-12 bf16 bound of
    test_tsfb_bf16_training_step_vs_oracle_f32 (eps*sqrt(10*12) = 1.2e-2, 2.5e-2 used) -- the fp16 rounding of the
    weights adds 2^-11/sqrt(3) per weight, an order below."""
    from lavila.models import models
    torch.manual_seed(0)
    with contextlib.redirect_stdout(io.StringIO()), warnings.catch_warnings():
        warnings.simplefilter('ignore')
        model = models.CLIP_OPENAI_TIMESFORMER_BASE(num_frames=4, project_embed_dim=256)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    w = O.procedural_weights(shapes, seed=17)
    model.load_state_dict(w)
    model.to(DEV).eval()
    model = model.half()
    video, tokens = O.synthetic_batch(3, 4, 224, seed=33)
    with warnings.catch_warnings(record=True) as rec, torch.no_grad():
        warnings.simplefilter('always')
        te = model.encode_text(tokens.to(DEV))
        ie = model.encode_image(video.to(DEV).half())
    left = [str(r.message) for r in rec if 'lavila_amd' in str(r.message) and 'CLIP weights' not in str(r.message)]
    assert not left, left                       # nothing fell off the hand-written kernels
    assert te.dtype == torch.float16 and ie.dtype == torch.float16
    torch.set_num_threads(min(32, torch.get_num_threads() or 1))
    with torch.no_grad():
        oo = O.clip_forward(video, tokens, w, 12, 8, norm_embed=True)
    e_img = _rel(O.l2_normalize(ie.float().cpu()), oo['image_embed'])
    e_txt = _rel(O.l2_normalize(te.float().cpu()), oo['text_embed'])
    print(f'[--use-half TSF-B] rel L2 image_embed {e_img:.2e} text_embed {e_txt:.2e}')
    assert e_img < 2.5e-2 and e_txt < 2.5e-2, (e_img, e_txt)


@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('B,Fr,img,P', [(2, 2, 32, 16), (2, 4, 224, 16), (2, 3, 42, 14), (2, 1, 24, 8)])
def test_patchify_reads_frame_major_clips_in_place(dt, B, Fr, img, P):
    """forward_features receives [B,F,C,H,W] (timesformer.py:345-348, narrator.py:74): the gather reads that layout in
    place (lvl_patchify frame_major=1) and must produce the bits of the BCFHW gather."""
    from lavila_amd import ops
    video = torch.randn(B, 3, Fr, img, img, generator=torch.Generator().manual_seed(5))
    a = ops.patchify(video.to(DEV), P, dt)
    b = ops.patchify(video.permute(0, 2, 1, 3, 4).contiguous().to(DEV), P, dt, frame_major=True)
    assert torch.equal(a, b)
    want = O.patchify(video, P).to(dt)
    assert torch.equal(a.cpu(), want)


def test_forward_features_frame_major_entry_equals_forward():
    fx, c, model = _tiny()
    model.to(DEV).eval()
    video, _ = synthetic_inputs(c, seed=fx['input_seed'])
    v = video.to(DEV)
    with torch.no_grad():
        a = model.visual(v)
        b = model.visual.forward_features(v.permute(0, 2, 1, 3, 4).contiguous())
        # a permuted VIEW (what a caller holding BCFHW data passes) is accepted as well
        c2 = model.visual.forward_features(v.permute(0, 2, 1, 3, 4))
    assert torch.equal(a, b) and torch.equal(a, c2)


def test_gated_and_drop_path_blocks_stay_on_own_gemms(monkeypatch):
    """timesformer.py:181-182 (tanh gating) and :192,196 (stochastic depth): those variants materialise the branch
    outputs -- still through lvl_linear_tn / lvl_linear_wgrad, not through nn.Linear (library GEMM)."""
    import torch.nn.functional as F
    from lavila.models.openai_model import QuickGELU
    from lavila.models.timesformer import SpaceTimeBlock
    from lavila_amd import ops
    torch.manual_seed(0)
    Fr, N, D, H, B = 2, 196, 768, 12, 2
    blk = SpaceTimeBlock(D, H, qkv_bias=True, act_layer=QuickGELU, time_init='rand', is_tanh_gating=True,
                         drop_path=0.5).to(DEV).train()
    with torch.no_grad():
        blk.alpha_timeattn.fill_(0.3)
        for p in blk.parameters():
            if p.ndim > 1:
                p.normal_(0, 0.02)
    x = torch.randn(B, 1 + Fr * N, D, device=DEV, dtype=torch.bfloat16, requires_grad=True)

    def no_library(*a, **k):
        raise AssertionError('nn.functional.linear reached from a SpaceTimeBlock in bf16')
    calls = []
    real = ops.linear_tn_raw
    monkeypatch.setattr(F, 'linear', no_library)
    monkeypatch.setattr(ops, 'linear_tn_raw', lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    with torch.autocast('cuda', dtype=torch.bfloat16):
        torch.manual_seed(3)
        y = blk(x, 'b (f n) d', '(b f) n d', 'b (f n) d', '(b n) f d', time_n=N, space_f=Fr)
    y.float().square().mean().backward()
    assert len(calls) >= 12                        # 6 forward + 6 input-gradient GEMMs
    assert torch.isfinite(x.grad).all() and blk.alpha_timeattn.grad is not None
    monkeypatch.undo()
    # numbers: same seeds on the oracle (gating and per-sample drop masks restated with torch ops in f32)
    w = {k: v.detach().float().cpu() for k, v in blk.state_dict().items()}
    blk.eval()
    with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16):
        y_eval = blk(x, 'b (f n) d', '(b f) n d', 'b (f n) d', '(b n) f d', time_n=N, space_f=Fr)
    # eval mode (drop path off, gating on) against the oracle block in f32: one block of bf16 roundings
    want = O.space_time_block(x.detach().float().cpu(), w, '', H, Fr, N)
    assert _rel(y_eval, want) < 2e-2, _rel(y_eval, want)


def test_training_forward_sees_param_data_writes():
    """ZeroRedundancyOptimizer / bucket views write parameters through `.data` (no version bump).
    A training forward must use the new values (the bf16 weight copies are re-cast per grad-enabled forward)."""
    from lavila.models.loss import CLIPLoss
    fx, c, model = _tiny()
    model.to(DEV).train()
    video, tokens = synthetic_inputs(c, seed=fx['input_seed'])
    v, t = video.to(DEV), tokens.to(DEV)

    def run():
        with torch.autocast('cuda', dtype=torch.bfloat16):
            out = model(v, t, norm_embed=True)
            return CLIPLoss()(out)['loss'].item(), out['image_embed'].detach().float().clone()
    l0, e0 = run()
    sd = {k: p.detach().clone() for k, p in model.named_parameters()}
    for k, p in model.named_parameters():
        if p.ndim == 2:
            p.data.mul_(1.5)                       # out-of-band write
    l1, e1 = run()
    assert (e1 - e0).abs().max() > 1e-3           # the forward saw the new weights
    fresh = build_model(c)
    fresh.load_state_dict({k: v for k, v in model.state_dict().items()})
    fresh.to(DEV).train()
    model2, model = model, fresh
    l2, e2 = run()                                 # a fresh model (empty cache) with the same values: identical bits
    assert l1 == l2 and torch.equal(e1, e2)
    del model2, sd


def test_hip_graph_replay_after_weight_update():
    """under capture the bf16 weight copies are cast INSIDE the graph, so a replay after the optimizer has
    changed the f32 masters computes with the updated weights (bit-equal to eager)."""
    from lavila.models.openai_model import QuickGELU
    from lavila.models.timesformer import SpaceTimeBlock
    from lavila_amd import ops
    torch.manual_seed(0)
    Fr, N, D, H, B = 2, 196, 768, 12, 2
    blk = SpaceTimeBlock(D, H, qkv_bias=True, act_layer=QuickGELU, time_init='rand').to(DEV)
    with torch.no_grad():
        for p in blk.parameters():
            if p.ndim > 1:
                p.normal_(0, 0.02)
    x_static = torch.randn(B, 1 + Fr * N, D, device=DEV, dtype=torch.bfloat16)

    def fwd():
        with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16):
            x1, y, b = blk.chain(x_static, None, None, Fr, N)
            return x1 + y + b.to(y.dtype)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fwd()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out_static = fwd()
    graph.replay()
    torch.cuda.synchronize()
    before = out_static.clone()
    with torch.no_grad():
        for p in blk.parameters():
            if p.ndim > 1:
                p.data.mul_(1.25)                 # what an optimizer step does to the masters
    graph.replay()
    torch.cuda.synchronize()
    after = out_static.clone()
    ops.invalidate_weight_cache()
    eager = fwd()
    torch.cuda.synchronize()
    assert not torch.equal(before, after)
    assert torch.equal(after, eager)


# ----------------------------------------------------------------------------------------------------------------------
# narrator, tower-side seam: VCLM_HF.encode_image on the HIP path
# ----------------------------------------------------------------------------------------------------------------------
def _narrator(fx):
    from lavila.models.openai_model import QuickGELU
    from lavila.models.timesformer import SpaceTimeTransformer
    from lavila_amd.narrator import VCLM_HF
    c = fx['config']
    with contextlib.redirect_stdout(io.StringIO()):
        vis = SpaceTimeTransformer(img_size=c['img'], patch_size=c['patch'], embed_dim=c['dim'], depth=c['depth'],
                                   num_heads=c['heads'], num_frames=c['frames'], time_init='zeros',
                                   attention_style='frozen-in-time', ln_pre=True, act_layer=QuickGELU)
    vis.head = vis.pre_logits = vis.fc = torch.nn.Identity()
    m = VCLM_HF(vision_width=c['dim'], vision_model=vis, text_width=c['text_width'], text_decoder=None,
                num_img_queries=c['queries'], dim_head=64, heads=c['pool_heads'])
    w = O.procedural_weights(fx['shapes'], seed=fx['weight_seed'])
    for k in fx['shapes']:
        if k.endswith('.beta'):
            w[k] = torch.zeros(fx['shapes'][k])
    m.load_state_dict(w, strict=True)
    return m.to(DEV).eval(), c


def test_narrator_encode_image_has_no_library_gemm_in_bf16(monkeypatch):
    """The pooling projections of VCLM_HF.encode_image (to_q / to_kv / to_out, coca.py:76-82) at the real widths: to_kv is
    [128 x 768], which the 256-column-panel kernel does not tile -- under no_grad it runs on lvl_linear_skinny's tiles."""
    import torch.nn.functional as F
    from lavila_amd.narrator import CrossAttention
    pool = CrossAttention(dim=768, context_dim=768, dim_head=64, heads=12, norm_context=True).to(DEV).eval()
    g = torch.Generator().manual_seed(0)
    queries = torch.randn(256, 768, generator=g).to(DEV)
    ctx = torch.randn(8, 785, 768, generator=g).to(DEV)
    with torch.no_grad():
        want = pool(queries, ctx)                                           # f32: library GEMMs
        monkeypatch.setattr(F, 'linear', lambda *a, **k: (_ for _ in ()).throw(AssertionError('library GEMM')))
        with torch.autocast('cuda', dtype=torch.bfloat16):
            got = pool(queries, ctx)
    assert got.dtype == torch.bfloat16
    assert _rel(got, want.cpu()) < 2e-2


@pytest.mark.parametrize('mode', ['f32', 'bf16', 'half'])
def test_narrator_encode_image_matches_reference(mode):
    """narrator.py:63-90 on the reference's own outputs (tests/golden/narrator_pool.pt): float32 within 1e-3, bf16
    autocast and the --use-half recipe of main_infer_narrator.py (docs/PRETRAIN.md:85-91) within the 2-block bf16 bound."""
    fx = load_golden('narrator_pool.pt')
    m, c = _narrator(fx)
    video, _ = O.synthetic_batch(c['batch'], c['frames'], c['img'], seed=fx['input_seed'])
    v = video.to(DEV)
    with torch.no_grad():
        if mode == 'f32':
            got = m.encode_image(v)
            assert got.dtype == torch.float32
            torch.testing.assert_close(got.cpu(), fx['image_tokens'], atol=1e-3, rtol=1e-3)
        elif mode == 'bf16':
            with torch.autocast('cuda', dtype=torch.bfloat16):
                got = m.encode_image(v)
            torch.testing.assert_close(got.float().cpu(), fx['image_tokens'], atol=4e-2, rtol=4e-2)
        else:
            got = m.half().encode_image(v.half())
            assert got.dtype == torch.float16
            torch.testing.assert_close(got.float().cpu(), fx['image_tokens'], atol=4e-2, rtol=4e-2)
    assert got.shape == (c['batch'], c['queries'], c['text_width'])


@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('B,NQ,H,T', [(2, 10, 3, 37), (3, 256, 12, 785), (1, 5, 1, 1), (2, 33, 2, 130)])
def test_mq_cross_attention_core(dt, B, NQ, H, T):
    """lvl_mq_cross_attn_fwd against oracle.mq_cross_attention_core (coca.py:104-120): per-sample and batch-shared
    queries, key counts that are not multiples of the 8-key blocks / 64-key chunks, one key."""
    from lavila_amd.narrator import mq_cross_attention
    g = torch.Generator().manual_seed(B * 1000 + T)
    q = torch.randn(B, NQ, H * 64, generator=g)
    kv = torch.randn(B, T, 128, generator=g)
    qd, kvd = q.to(DEV).to(dt), kv.to(DEV).to(dt)
    want = O.mq_cross_attention_core(qd.float().cpu(), kvd.float().cpu(), H)
    got = mq_cross_attention(qd, kvd, H)
    tol = dict(atol=2e-5, rtol=1e-4) if dt == torch.float32 else dict(atol=2e-2, rtol=2e-2)
    torch.testing.assert_close(got.float().cpu(), want, **tol)
    shared = mq_cross_attention(qd[0], kvd, H)                       # [NQ, H*64]: the same queries for every sample
    want_s = O.mq_cross_attention_core(qd[:1].float().cpu().expand(B, -1, -1), kvd.float().cpu(), H)
    torch.testing.assert_close(shared.float().cpu(), want_s, **tol)


def test_narrator_general_cross_attention_and_inference_only():
    from lavila_amd._cabi import HipExtensionError
    fx = load_golden('narrator_pool.pt')
    m, c = _narrator(fx)
    g = torch.Generator().manual_seed(fx['pool_general_seed'])
    xq = torch.randn(2, 10, c['text_width'], generator=g)
    ctx = torch.randn(2, 37, c['dim'], generator=g)
    with torch.no_grad():
        got = m.img_attn_pool(xq.to(DEV), ctx.to(DEV))
    torch.testing.assert_close(got.cpu(), fx['pool_general'], atol=1e-3, rtol=1e-3)
    out = m.img_attn_pool(xq.to(DEV).requires_grad_(True), ctx.to(DEV))
    with pytest.raises(HipExtensionError):
        out.sum().backward()


@pytest.mark.parametrize('N,K', [(768, 768), (2304, 768), (100, 72), (65, 130), (512, 2048)])
def test_cast_transpose(N, K):
    """lvl_cast_transpose: f32 master [N,K] -> bf16 copy and bf16 transposed copy, bit-equal to torch's rounding."""
    from lavila_amd import _cabi as C
    src = torch.randn(N, K, device=DEV, generator=torch.Generator(device=DEV).manual_seed(N + K))
    w = torch.empty(N, K, dtype=torch.bfloat16, device=DEV)
    wt = torch.empty(K, N, dtype=torch.bfloat16, device=DEV)
    C.check(C.lib().lvl_cast_transpose(C.ptr(src), C.ptr(w), C.ptr(wt), N, K, C.stream_ptr()), 'lvl_cast_transpose')
    assert torch.equal(w, src.bfloat16()) and torch.equal(wt, src.bfloat16().t().contiguous())
