"""CPU (-m "not gpu"): the drop-in boundary -- import paths, constructor kwargs, state_dict names, C-ABI
exports, loud failure without a GPU, checkpoint helpers vs the reference."""
import ctypes
import os
import re

import pytest
import torch

from conftest import ROOT, load_golden
from helpers import build_model
from oracle.ref_import import reference_available

MODELS = ['tiny_p16', 'tiny_p14_gated', 'config1_tsfb_112']


def test_cabi_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, 'include', 'lavila_hip.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    declared = set(re.findall(r'\b(lvl_[a-z0-9_]+)\s*\(', hdr))
    assert len(declared) >= 15
    from lavila_amd import _cabi
    assert declared == set(_cabi.SIGNATURES), declared ^ set(_cabi.SIGNATURES)
    lib = ctypes.CDLL(_cabi.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert b'gfx950' in _cabi.lib().lvl_version()
    assert _cabi.lib().lvl_workspace_floats(b'layernorm_bwd', 10, 768) > 0
    # ... and nothing else: the library is built with -fvisibility=hidden, its C++ internals (launch helpers, lvl_fail,
    # kernel host stubs) stay inside (34 mangled symbols were exported)
    import shutil
    import subprocess
    nm = shutil.which('nm') or '/opt/rocm/lib/llvm/bin/llvm-nm'
    if os.path.exists(nm):
        out = subprocess.run([nm, '-D', '--defined-only', _cabi.LIB_PATH], capture_output=True, text=True, check=True).stdout
        exported = {l.split()[-1] for l in out.splitlines() if len(l.split()) >= 3 and l.split()[-2] in 'TtDdBbRr'}
        funcs = {l.split()[-1] for l in out.splitlines() if len(l.split()) >= 3 and l.split()[-2] == 'T'}
        assert funcs == declared, funcs ^ declared
        assert not [e for e in exported if e.startswith('_Z') and 'lvl' in e], 'mangled internals exported'


def test_workspace_queries_are_host_only_and_shape_aware():
    """lvl_workspace_floats is pure host code: every op the header names answers without a GPU, unknown ops and
    shapes the weight-gradient kernel cannot tile answer -1 (the caller then keeps the library GEMM)."""
    from lavila_amd import _cabi
    ws = _cabi.lib().lvl_workspace_floats
    for op, r, c in ((b'layernorm_bwd', 10, 768), (b'bias_quickgelu_bwd', 10, 3072), (b'divided_attn_fwd', 24, 785),
                     (b'divided_attn_bwd', 24, 785), (b'causal_attn_bwd', 16, 77), (b'qkv_bias_grad', 1000, 768)):
        assert ws(op, r, c) > 0, op
    assert ws(b'no_such_op', 1, 1) == -1
    # weight-gradient tilings: (N, K) multiples of 192/288/384 (6x6 wave tiles) or 128/256 (8x4 wave tiles)
    for n, k in ((768, 768), (2304, 768), (3072, 768), (768, 3072), (1024, 1024), (4096, 1024), (1536, 512), (256, 128)):
        assert ws(b"linear_wgrad", n, k) >= n * k * 4, (n, k)          # S >= 4 f32 partial tiles (S * tiles ~ 256)
    for n, k in ((200, 200), (384, 128), (100, 768)):
        assert ws(b'linear_wgrad', n, k) == -1, (n, k)


@pytest.mark.parametrize('name', MODELS)
def test_state_dict_names_and_shapes_match_reference(name):
    fx = load_golden(f'model_{name}.pt')
    m = build_model(fx['config'])
    sd = m.state_dict()
    assert list(sd.keys()) == fx['state_dict_keys']
    assert [k for k, _ in m.named_parameters()] == fx['param_names']
    assert {k: tuple(v.shape) for k, v in sd.items()} == fx['shapes']


def test_named_constructor_swallows_driver_kwargs_and_matches_survey_counts():
    from lavila.models import models
    m = models.CLIP_OPENAI_TIMESFORMER_BASE(
        pretrained=False, pretrained2d=True, text_use_cls_token=False, project_embed_dim=256, gated_xattn=False,
        random_init_gpt2=False, timesformer_gated_xattn=False, timesformer_freeze_space=False, freeze_lm_vclm=False,
        freeze_visual_vclm=False, freeze_visual_vclm_temporal=False, num_frames=4, drop_path_rate=0.0,
        temperature_init=0.07)
    assert len(m.state_dict()) == 375                                  # SURVEY.md 8b [probed]
    assert abs(sum(p.numel() for p in m.parameters()) / 1e6 - 177.66) < 0.01
    # shipped init: temporal attention starts as an exact no-op (timesformer.py:97-103)
    ta = m.visual.blocks[0].timeattn
    assert float(ta.qkv.weight.abs().max()) == 0 and float(ta.proj.weight.min()) == 1
    assert m.visual.patch_embed.proj.bias is None                     # ln_pre=True -> bias=False
    assert models.get_metric_names('CLIP_OPENAI_TIMESFORMER_BASE') == ['loss', 'clip_loss', 'clip_acc']
    args = type('A', (), dict(contrastive_use_vissl=True, rank=0, world_size=1))
    crit = models.get_loss('CLIP_OPENAI_TIMESFORMER_BASE', args)
    assert isinstance(crit, models.loss.CLIPLoss) and crit.use_vissl and crit.state_dict() == {}
    with pytest.raises(NotImplementedError):
        models.get_loss('VCLM_OPENAI_TIMESFORMER_BASE_GPT2', args)


@pytest.mark.skipif(not reference_available(), reason='reference only exists in the build container')
@pytest.mark.parametrize('ctor,frames', [('CLIP_OPENAI_TIMESFORMER_LARGE', 4), ('CLIP_OPENAI_TIMESFORMER_LARGE_336PX', 16)])
def test_large_constructors_match_reference_state_dict(ctor, frames):
    """models.py:374-491: TSF-L/14 at 224 (N=256) and 336 (N=576): parameter names, shapes, order, and the freeze-space
    error behaviour when no CLIP weights are present."""
    import contextlib
    import io
    from lavila.models import models
    from oracle.ref_import import load_reference
    import torch.nn as nn
    ref = load_reference()
    with contextlib.redirect_stdout(io.StringIO()):
        ours = getattr(models, ctor)(num_frames=frames, pretrained=False, project_embed_dim=256)
        size = 336 if '336' in ctor else 224
        rv = ref.timesformer.SpaceTimeTransformer(
            img_size=size, patch_size=14, embed_dim=1024, depth=24, num_heads=16, num_frames=frames, time_init='zeros',
            attention_style='frozen-in-time', ln_pre=True, act_layer=ref.openai_model.QuickGELU, is_tanh_gating=False)
        rv.head = nn.Identity()
        rv.pre_logits = nn.Identity()
        rv.fc = nn.Identity()
        theirs = ref.models.CLIP(embed_dim=256, vision_width=1024, vision_model=rv, context_length=77, vocab_size=49408,
                                 transformer_width=768, transformer_heads=12, transformer_layers=12, tempearture_init=0.07)
    a = {k: tuple(v.shape) for k, v in ours.state_dict().items()}
    b = {k: tuple(v.shape) for k, v in theirs.state_dict().items()}
    assert list(a) == list(b) and a == b
    assert ours.visual.patches_per_frame == (size // 14) ** 2
    with pytest.raises(RuntimeError):      # nothing would be frozen without the pretrained weights: say so
        with contextlib.redirect_stdout(io.StringIO()):
            getattr(models, ctor)(num_frames=frames, timesformer_freeze_space=True)


def test_no_cpu_fallback():
    from lavila_amd._cabi import HipExtensionError
    fx = load_golden('model_tiny_p16.pt')
    m = build_model(fx['config'])
    with pytest.raises(HipExtensionError):
        m(torch.randn(2, 3, 2, 32, 32), torch.zeros(2, 77, dtype=torch.long))
    with pytest.raises(HipExtensionError):
        m.visual.norm(torch.randn(4, 128))
    from lavila.models.loss import CLIPLoss
    with pytest.raises(HipExtensionError):
        CLIPLoss()({'image_embed': torch.randn(4, 8), 'text_embed': torch.randn(4, 8),
                    'logit_scale': torch.tensor(10.0)})


def test_reference_error_behaviour():
    from lavila.models.timesformer import SpaceTimeTransformer
    with pytest.raises(NotImplementedError):
        SpaceTimeTransformer(hybrid_backbone=object())
    from lavila.models.loss import CLIPLoss
    with pytest.raises(RuntimeError):
        CLIPLoss(world_size=2)({'image_embed': torch.randn(2, 8), 'text_embed': torch.randn(2, 8),
                                'logit_scale': torch.tensor(1.0)})


@pytest.mark.skipif(not reference_available(), reason='reference only exists in the build container')
def test_checkpoint_helpers_match_reference():
    from oracle.ref_import import load_reference
    from lavila.models.utils import inflate_positional_embeds, remap_keys
    ref = load_reference()
    g = torch.Generator().manual_seed(3)
    clip = {'class_embedding': torch.randn(8, generator=g), 'positional_embedding': torch.randn(5, 8, generator=g),
            'conv1.weight': torch.randn(8, 3, 2, 2, generator=g), 'ln_pre.weight': torch.randn(8, generator=g),
            'ln_pre.bias': torch.randn(8, generator=g), 'ln_post.weight': torch.randn(8, generator=g),
            'ln_post.bias': torch.randn(8, generator=g), 'proj': torch.randn(8, 4, generator=g)}
    for layer in range(2):
        for leaf, shp in [('attn.in_proj_weight', (24, 8)), ('attn.in_proj_bias', (24,)),
                          ('attn.out_proj.weight', (8, 8)), ('attn.out_proj.bias', (8,)), ('ln_1.weight', (8,)),
                          ('ln_1.bias', (8,)), ('mlp.c_fc.weight', (32, 8)), ('mlp.c_fc.bias', (32,)),
                          ('mlp.c_proj.weight', (8, 32)), ('mlp.c_proj.bias', (8,)), ('ln_2.weight', (8,)),
                          ('ln_2.bias', (8,))]:
            clip[f'transformer.resblocks.{layer}.{leaf}'] = torch.randn(*shp, generator=g)
    ours = remap_keys(dict(clip), transformer_layers=2)
    theirs = ref.utils.remap_keys(dict(clip), transformer_layers=2)
    assert list(ours.keys()) == list(theirs.keys())
    for k in ours:
        assert torch.equal(ours[k], theirs[k]), k
    for have, want, fix in [(4, 8, 'bilinear'), (8, 4, 'bilinear'), (4, 6, 'zeros'), (4, 7, 'interp'), (4, 4, 'bilinear')]:
        cur = {'visual.temporal_embed': torch.zeros(1, want, 8), 'visual.pos_embed': torch.zeros(1, 5, 8)}
        new = {'visual.temporal_embed': torch.randn(1, have, 8, generator=g), 'visual.pos_embed': torch.zeros(1, 5, 8)}
        a = inflate_positional_embeds(cur, dict(new), num_frames=want, load_temporal_fix=fix)
        b = ref.utils.inflate_positional_embeds(cur, dict(new), num_frames=want, load_temporal_fix=fix)
        assert torch.equal(a['visual.temporal_embed'], b['visual.temporal_embed'])
    with pytest.raises(NotImplementedError):
        inflate_positional_embeds({'visual.pos_embed': torch.zeros(1, 5, 8)}, {'visual.pos_embed': torch.zeros(1, 10, 8)})


@pytest.mark.skipif(not reference_available(), reason='reference only exists in the build container')
def test_drop_in_coexists_with_reference_tree():
    """With this repo ahead of the reference on sys.path, the hot-path modules resolve to lavila_amd while every
    other reference module (meters, schedulers, ...) still comes from the reference tree."""
    import subprocess
    import sys
    from oracle.ref_import import REFERENCE_ROOT
    code = (
        "import sys; sys.path[:0] = [%r, %r]\n"
        "import lavila.models.models as m, lavila.models.loss as l, lavila.models.timesformer as t\n"
        "import lavila.utils.meter as meter, lavila.utils.scheduler as sch, lavila.utils.distributed as d\n"
        "assert m.__name__ == 'lavila_amd.models' and l.__name__ == 'lavila_amd.loss' and t.__name__ == 'lavila_amd.timesformer'\n"
        "assert meter.__file__.startswith(%r) and sch.__file__.startswith(%r), meter.__file__\n"
        "assert d.__file__.startswith(%r), d.__file__      # host-side process-group glue stays the reference's\n"
        "assert hasattr(m, 'CLIP_OPENAI_TIMESFORMER_BASE') and hasattr(m.loss, 'CLIPLoss')\n"
        "print('ok')\n" % (ROOT, REFERENCE_ROOT, REFERENCE_ROOT, REFERENCE_ROOT, REFERENCE_ROOT))
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and 'ok' in out.stdout, out.stderr[-2000:]


@pytest.mark.skipif(not reference_available(), reason='reference only exists in the build container')
@pytest.mark.parametrize('embed,freeze', [(512, True), (256, False)])
def test_pretrained_clip_import_matches_reference_constructor(tmp_path, monkeypatch, embed, freeze):
    """models.py:329-370: the named constructor imports OpenAI-CLIP weights (vision tower through remap_keys with
    strict=False, text tower / embeddings / ln_final verbatim, the two projections + logit_scale only when
    project_embed_dim equals CLIP's 512) and optionally freezes what it imported. No network here: a seeded-random
    `openai_model.CLIP` of the ViT-B/16 shape stands in for the download on BOTH sides -- the reference constructor
    gets it through a monkey-patched `load_openai_clip`, ours reads the same state_dict from LAVILA_CLIP_WEIGHTS_DIR --
    and every tensor of the two resulting models must be equal, requires_grad flags included."""
    import contextlib
    import io
    from lavila.models import models
    from oracle.ref_import import load_reference
    ref = load_reference()
    torch.manual_seed(11)
    with contextlib.redirect_stdout(io.StringIO()):
        clip = ref.openai_model.CLIP(embed_dim=512, image_resolution=224, vision_layers=12, vision_width=768,
                                     vision_patch_size=16, context_length=77, vocab_size=49408, transformer_width=512,
                                     transformer_heads=8, transformer_layers=12)
    with torch.no_grad():       # make every imported tensor distinguishable from any default initialisation
        g = torch.Generator().manual_seed(12)
        for p in clip.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * 0.05)
    torch.save(clip.state_dict(), tmp_path / 'ViT-B-16.pt')
    monkeypatch.setenv('LAVILA_CLIP_WEIGHTS_DIR', str(tmp_path))
    monkeypatch.setattr(ref.models, 'load_openai_clip', lambda name, device='cpu': (clip, None))
    kw = dict(num_frames=4, timesformer_freeze_space=freeze, project_embed_dim=embed, temperature_init=0.07)
    with contextlib.redirect_stdout(io.StringIO()):
        theirs = ref.models.CLIP_OPENAI_TIMESFORMER_BASE(**kw)
        ours = models.CLIP_OPENAI_TIMESFORMER_BASE(**kw)
    a, b = ours.state_dict(), theirs.state_dict()
    assert list(a) == list(b)
    random_init = set() if embed == 512 else {'image_projection', 'text_projection'}    # normal_() draws, not imports
    for k in a:
        if k in random_init:
            assert a[k].shape == b[k].shape
            continue
        assert torch.equal(a[k], b[k]), k
    if embed == 512:
        assert torch.equal(a['logit_scale'], clip.logit_scale.data) and torch.equal(a['image_projection'], clip.visual.proj.data)
    else:
        assert abs(float(a['logit_scale']) - float(torch.log(torch.tensor(1 / 0.07)))) < 1e-6
    assert torch.equal(a['visual.blocks.3.attn.qkv.weight'], clip.visual.transformer.resblocks[3].attn.in_proj_weight.data)
    assert torch.equal(a['transformer.resblocks.5.mlp.c_fc.weight'], clip.transformer.resblocks[5].mlp.c_fc.weight.data)
    flags_a = {k: p.requires_grad for k, p in ours.named_parameters()}
    flags_b = {k: p.requires_grad for k, p in theirs.named_parameters()}
    assert flags_a == flags_b
    if freeze:
        assert not flags_a['visual.blocks.0.attn.qkv.weight'] and flags_a['visual.blocks.0.timeattn.qkv.weight'] \
            and flags_a['visual.cls_token'] and flags_a['visual.temporal_embed']
    # a missing file is an error, as in the reference (openai_clip.py:128)
    monkeypatch.setenv('LAVILA_CLIP_WEIGHTS_DIR', str(tmp_path / 'nowhere'))
    with pytest.raises(RuntimeError):
        with contextlib.redirect_stdout(io.StringIO()):
            models.CLIP_OPENAI_TIMESFORMER_BASE(num_frames=4)


def test_weight_cache_sees_out_of_band_parameter_writes():
    """`param.data` writes (ZeroRedundancyOptimizer's shard broadcast, bucket views) do not bump the
    version counter the bf16 weight-copy cache keys on. The cache generation -- bumped by every optimizer step (global
    post-step hook) and by every grad-enabled model forward -- makes them visible. fp16 weights take the torch cast
    path, so the cache logic itself runs without a GPU."""
    from lavila_amd import ops
    p = torch.nn.Parameter(torch.randn(8, 4).half())
    w0, wt0 = ops.weight_copies(p)
    assert ops.weight_copies(p)[0] is w0                          # cached
    p.data.add_(1.0)                                              # out-of-band write: version unchanged
    assert ops.weight_copies(p)[0] is w0                          # ... which is exactly the hazard
    torch.optim.SGD([p], lr=0.0).step()                           # any optimizer step invalidates
    w1, wt1 = ops.weight_copies(p)
    assert w1 is not w0 and torch.equal(w1.float(), p.detach().to(torch.bfloat16).float())
    assert torch.equal(wt1, w1.t())
    p.data.add_(1.0)
    with torch.no_grad():
        ops.training_forward_begins()                             # inference forwards keep the cache
    assert ops.weight_copies(p)[0] is w1
    ops.training_forward_begins()                                 # a training forward re-casts
    w2 = ops.weight_copies(p)[0]
    assert w2 is not w1 and torch.equal(w2.float(), p.detach().to(torch.bfloat16).float())
    p.data.add_(1.0)
    ops.invalidate_weight_cache()                                 # the explicit entry point
    assert torch.equal(ops.weight_copies(p)[0].float(), p.detach().to(torch.bfloat16).float())
    with torch.no_grad():
        p.add_(1.0)                                               # in-band write: the version counter alone suffices
    assert torch.equal(ops.weight_copies(p)[0].float(), p.detach().to(torch.bfloat16).float())


def _isa(tmp_path, name):
    import shutil
    import subprocess
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        pytest.skip('hipcc not available')
    src = os.path.join(ROOT, 'lavila_amd', 'csrc', name + '.hip')
    out = tmp_path / (name + '.s')
    from lavila_amd.build import EXTRA_FLAGS          # the per-file flags of the shipped build
    subprocess.run([hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-fno-gpu-rdc', '-S',
                    '--cuda-device-only', *EXTRA_FLAGS.get(name + '.hip', []), src, '-o', str(out)], check=True,
                   capture_output=True, timeout=900)
    return open(out).read().split('\n')


def _kernel_bodies(lines, stem):
    starts = [i for i, l in enumerate(lines) if l.startswith('_ZN') and stem in l and l.split(':')[0].isidentifier()]
    for a in starts:
        b = next(i for i in range(a, len(lines)) if lines[i].startswith('.Lfunc_end'))      # (s_endpgm may occur early)
        yield lines[a], [l.strip() for l in lines[a:b]]


def _check_parked_reply_registers(name, body, n_defs):
    """Registers that receive a memory reply LATER than the instruction naming them (`pend` in the GEMM): destinations of
    the returning atomics issued from the exec-narrowed asm blocks. Static check on the control-flow graph of the ISA:
    on every path from such a request to (a) the mailbox `ds_write_b32 vX, vN` that consumes it or (b) a full
    `s_waitcnt vmcnt(0)`, no other instruction may touch the register -- no copy, spill, reload or reuse while the
    reply can still be in flight."""
    body = [l.split(';')[0].strip() for l in body]
    ins = [l for l in body if l and (not l.startswith('.') or re.fullmatch(r'\.LBB\d+_\d+:', l))]
    labels = {l[:-1]: i for i, l in enumerate(ins) if l.endswith(':')}
    defs = [(i, re.search(r' (v\d+),', l).group(1)) for i, l in enumerate(ins)
            if i > 0 and ins[i - 1] == 's_mov_b64 exec, 1' and re.match(r'global_atomic_add v\d+, v\d+, v\d+, s\[\d+:\d+\] sc0$', l)]
    assert len(defs) == n_defs, (name, defs)

    def touches(l, reg):
        n = int(reg[1:])
        return re.search(r'\b' + reg + r'\b', l) is not None or \
            any(int(m.group(1)) <= n <= int(m.group(2)) for m in re.finditer(r'v\[(\d+):(\d+)\]', l))

    def succ(i):
        l = ins[i]
        if l.startswith('s_endpgm'):
            return []
        if l.startswith('s_branch'):
            return [labels[l.split()[-1]]]
        out = [i + 1] if i + 1 < len(ins) else []
        if l.startswith('s_cbranch'):
            out.append(labels[l.split()[-1]])
        return out
    for d, reg in defs:
        seen, todo, published = set(), succ(d), 0
        while todo:
            i = todo.pop()
            if i in seen:
                continue
            seen.add(i)
            l = ins[i]
            if re.fullmatch(r'ds_write_b32 v\d+, ' + reg, l):
                published += 1
                continue                                   # consumed: the window ends here
            if l == 's_waitcnt vmcnt(0)':
                published += 1
                continue                                   # every reply has been delivered
            nxt = [k for k in range(i, min(i + 8, len(ins))) if ins[k].startswith('global_atomic_add ' + reg + ',')
                   and ins[k - 1] == 's_mov_b64 exec, 1']
            if nxt and (nxt[0] == i or re.fullmatch(r'v_mov_b32_e32 ' + reg + r', [01]', l)):
                # the next request into the same register (replies return in issue order; the compiler may load the
                # request's constant operand into the very register that will receive the reply); its own window is
                # checked separately. (The walk is path-insensitive: it also follows the nb < 3 exit of the K loop.)
                continue
            assert not touches(l, reg), f'{name[:70]}: {reg} (reply in flight since instruction {d}) touched by `{l}`'
            todo.extend(succ(i))
        assert published >= 1, (name, d, reg)


def test_gemm_tile_counter_reply_register_is_untouched_in_isa(tmp_path):
    """lvl_linear_tn's dynamic tile scheduler and lvl_linear_wgrad's chunk claims park the reply of a returning atomic in
    a VGPR that the memory system writes LATER than the instruction that names it (`pend` in csrc/gemm_tn_mfma.hip and
    csrc/wgrad_mfma.hip). That is only sound if the compiler never copies, spills or reuses that register between the
    request and the LDS publish behind the covering vmcnt wait. Checked where it can be checked without a GPU: in the
    gfx950 ISA of every instantiation (see _check_parked_reply_registers); the kernels of the benched configuration must
    also be free of scratch spills (a spill reload is a vector-memory operation inside the counted vmcnt schedule)."""
    lines = _isa(tmp_path, 'gemm_tn_mfma')
    bodies = list(_kernel_bodies(lines, 'gemm_tn_kernel'))
    assert len(bodies) == 10                # 6 bf16 epilogues + 4 f32-class ones (Lb1E: float32 results)
    for name, body in bodies:
        if 'Lb0E' in name:
            # no spill traffic between the first and the last MFMA (K loop + epilogues of every tile): a reload there is a
            # vector-memory operation the counted vmcnt waits do not know, and the compiler drains vmcnt to 0 for it
            mf = [i for i, l in enumerate(body) if l.startswith('v_mfma')]
            assert not any('scratch_' in l for l in body[mf[0]:mf[-1]]), f'VGPR spills inside the tile loop of {name}'
        _check_parked_reply_registers(name, body, 3)           # first hand-out, tile 1, the per-tile pull
    lines = _isa(tmp_path, 'wgrad_mfma')
    bodies = list(_kernel_bodies(lines, 'wgrad_kernel'))
    assert len(bodies) == 16
    for name, body in bodies:
        if 'Lb0E' in name and ('ILi4ELi2ELi6ELi6E' in name or 'ILi2ELi4ELi6ELi6E' in name or 'ILi2ELi4ELi8ELi4E' in name):
            assert not any('scratch_' in l for l in body), name      # the TSF-B / TSF-L / text shapes of the towers
        # the chunk claims are synchronous: request, full wait and mailbox write sit in ONE asm block (nothing parked);
        # the dbias instantiations keep the static plan and contain no claim code at all
        reqs = [i for i, l in enumerate(body) if i > 0 and body[i - 1] == 's_mov_b64 exec, 1' and l.endswith((' sc0', ' sc1'))]
        assert len(reqs) == (3 if 'Lb0E' in name else 0), (name, reqs)
        for i in reqs:
            reg = re.search(r' (v\d+),', body[i]).group(1)
            assert body[i + 1] == 's_waitcnt vmcnt(0)' and re.fullmatch(r'ds_write_b32 v\d+, ' + reg, body[i + 2]), body[i:i + 3]


def test_exact_width_layernorm_loops_have_only_counted_waits_in_isa(tmp_path):
    """ln_fwd_exact_kernel / ln_bwd_exact_kernel (csrc/layernorm.hip) exist because ONE conditional vector-memory instruction
    in a row loop makes the compiler's wait insertion fall back to `s_waitcnt vmcnt(0)` there (the software-prefetched next
    row is then waited for as well, and every store acknowledgement sits on the row's critical path). Checked in the gfx950
    ISA of the benched instantiations: the row loop (the backward-branching block that loads and stores) contains no
    vmcnt(0), no branch around a memory instruction, and at least one counted wait."""
    lines = _isa(tmp_path, 'layernorm')
    seen = 0
    for stem in ('ln_fwd_exact_kernel', 'ln_bwd_exact_kernel'):
        for name, body in _kernel_bodies(lines, stem):
            if '6bf16_tLi3ELi4E' not in name:      # 768 columns: TSF-B, the benched tower (other widths: the compiler may
                continue                           # rotate the loop so that its last counted wait is an exact vmcnt(0))
            labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r'(\.LBB\d+_\d+):', l)] if m}
            loops = []
            for i, l in enumerate(body):
                m = re.match(r's_cbranch_\w+ (\.LBB\d+_\d+)', l)
                if m and m.group(1) in labels and labels[m.group(1)] < i:
                    seg = body[labels[m.group(1)]:i]
                    if any(x.startswith('global_load') for x in seg) and any(x.startswith('global_store') for x in seg):
                        loops.append(seg)
            assert len(loops) == 1, (name, len(loops))
            seg = loops[0]
            waits = [x for x in seg if x.startswith('s_waitcnt') and 'vmcnt' in x]
            assert waits and not any('vmcnt(0)' in x for x in waits), (name, waits)
            assert not any(x.startswith(('s_cbranch', 's_branch')) for x in seg), name       # straight-line row loop
            seen += 1
    assert seen >= 7          # 3 forward + 4 backward operand combinations


def test_narrator_seam_state_dict_matches_reference_names():
    """lavila_amd.narrator.VCLM_HF owns `visual.*`, `img_queries`, `img_attn_pool.*`, `img_attn_pool_norm.*` under the
    reference's names (narrator.py:44-49, coca.py:27-31,76-82), beta buffers included, so those entries of a VCLM_*
    checkpoint load unchanged; without a decoder module forward and the beam searches say so."""
    import contextlib
    import io
    from lavila.models.openai_model import QuickGELU
    from lavila.models.timesformer import SpaceTimeTransformer
    from lavila_amd.narrator import VCLM_HF, CrossAttention
    fx = load_golden('narrator_pool.pt')
    c = fx['config']
    with contextlib.redirect_stdout(io.StringIO()):
        vis = SpaceTimeTransformer(img_size=c['img'], patch_size=c['patch'], embed_dim=c['dim'], depth=c['depth'],
                                   num_heads=c['heads'], num_frames=c['frames'], time_init='zeros',
                                   attention_style='frozen-in-time', ln_pre=True, act_layer=QuickGELU)
    vis.head = vis.pre_logits = vis.fc = torch.nn.Identity()
    m = VCLM_HF(vision_width=c['dim'], vision_model=vis, text_width=c['text_width'], text_decoder=None,
                num_img_queries=c['queries'], dim_head=64, heads=c['pool_heads'])
    sd = m.state_dict()
    assert list(sd.keys()) == fx['state_dict_keys']
    assert {k: tuple(v.shape) for k, v in sd.items()} == fx['shapes']
    assert [k for k, _ in m.named_buffers()] == ['img_attn_pool.norm.beta', 'img_attn_pool.context_norm.beta',
                                                 'img_attn_pool_norm.beta']
    with pytest.raises(NotImplementedError):
        m.beam_sample(None, None)
    with pytest.raises(NotImplementedError):
        m(torch.zeros(1, 3, 2, 32, 32), torch.zeros(1, 8, dtype=torch.long))
    with pytest.raises(NotImplementedError):
        CrossAttention(64, parallel_ff=True)


def test_no_memset_or_memcpy_nodes_in_the_c_abi():
    """Every entry point may run under hipGraph capture (GraphedTrainStep). A hipMemsetAsync there becomes a memset NODE, and
    a replayed memset node is not reliable on this ROCm build (round 5: its fill pattern came from recycled memory --
    csrc/common.h, lvl_zero_f32); a hipMemcpyAsync from host memory would bake a host pointer into the graph. Fills and
    copies inside the library are kernels."""
    import glob
    import re
    bad = []
    for path in sorted(glob.glob(os.path.join(ROOT, 'lavila_amd', 'csrc', '*'))):
        for n, line in enumerate(open(path), 1):
            code = line.split('//')[0]
            if re.search(r'\bhipMem(set|cpy)\w*\s*\(', code):
                bad.append(f'{os.path.basename(path)}:{n}: {line.strip()}')
    assert not bad, bad


def test_drop_in_import_reserves_hardware_queues_for_ranks():
    """VERDICT r5 missing 5: an unmodified `torchrun main_pretrain.py` must not lose the tower overlap to RCCL's streams.
    Importing the drop-in with WORLD_SIZE > 1 (and the HIP runtime not yet up) exports GPU_MAX_HW_QUEUES=8; a value the
    user set is kept; a single process leaves the environment alone (main_pretrain.py:28 imports lavila.models before
    :151-183 touch the device)."""
    import subprocess
    import sys
    code = 'import os, lavila.models.models as m, lavila_amd; print(os.environ.get("GPU_MAX_HW_QUEUES"), lavila_amd.HW_QUEUES_RESERVED)'

    def run(**env):
        e = {k: v for k, v in os.environ.items() if k not in ('GPU_MAX_HW_QUEUES', 'WORLD_SIZE', 'LAVILA_HW_QUEUES')}
        e.update(env)
        e['PYTHONPATH'] = ROOT + os.pathsep + e.get('PYTHONPATH', '')
        return subprocess.run([sys.executable, '-c', code], env=e, capture_output=True, text=True, check=True).stdout.split()[-2:]

    assert run(WORLD_SIZE='8') == ['8', 'True']
    assert run(WORLD_SIZE='8', GPU_MAX_HW_QUEUES='2') == ['2', 'None']
    assert run(WORLD_SIZE='1') == ['None', 'None']
    assert run() == ['None', 'None']
    assert run(WORLD_SIZE='8', LAVILA_HW_QUEUES='0') == ['None', 'None']
