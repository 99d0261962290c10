"""Model zoo / API of the dual encoder, mirroring `lavila/models/models.py` of the reference:
`CLIP` (:75-173), `get_loss` (:293-304), `get_metric_names` (:307-313) and the named constructors
`CLIP_OPENAI_TIMESFORMER_{BASE,LARGE,LARGE_336PX}` (:316-491) -- same names, kwargs (unknown kwargs are
swallowed exactly as the reference's **kwargs do), output dict keys and state_dict keys, so that
main_pretrain.py / eval_zeroshot.py drive it unchanged. Built: the dual-encoder pretraining path
and, for inference, the narrator on TimeSformer towers (`VCLM_OPENAI_TIMESFORMER_*`, :887-1198; lavila_amd.narrator +
lavila_amd.gpt2_gated). The per-frame ViT narrators (`VCLM_OPENAI_VIT*`), DistilBERT and fine-tuning heads are absent.
"""
import contextlib
import os
import threading
import weakref

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import loss, ops
from .gpt2_gated import GPT2LMHeadModel as GatedGPT2LMHeadModel
from .gpt2_gated import augment_gpt2_config, gpt2_config
from .narrator import VCLM_HF
from .openai_model import QuickGELU, Transformer
from .timesformer import LayerNorm, SpaceTimeTransformer
from .utils import remap_keys, rsetattr  # noqa: F401  (re-exported like the reference)


@contextlib.contextmanager
def _amp_region():
    """The reference drivers wrap forward+loss in torch.cuda.amp.autocast() = fp16 (main_pretrain.py:490).
    The kernels are f32/bf16: an active fp16 autocast region is re-entered as bf16 (GradScaler stays
    harmless: bf16 has f32's exponent range)."""
    if torch.is_autocast_enabled() and torch.get_autocast_dtype('cuda') == torch.float16:
        with torch.autocast('cuda', dtype=torch.bfloat16):
            yield
    else:
        yield


def _half_out(x, *given):
    """`model.half()` / `images.half()` callers (eval_zeroshot.py:212-213,256-261,293-304,340-354; no autocast there)
    get fp16 back; inside, fp16 parameters and activations are computed in bf16 (ops.lowp, ops.weight_copies)."""
    if not torch.is_autocast_enabled() and x.dtype != torch.float16 and \
            any(g.dtype == torch.float16 for g in given):
        return x.to(torch.float16)
    return x


_TEXT_STREAM = os.environ.get('LAVILA_TEXT_STREAM', '1') != '0'
_TEXT_TRIM = os.environ.get('LAVILA_TEXT_TRIM', '1') != '0'
_TEXT_AFTER_BLOCK = 1          # video blocks enqueued before the text tower's caption-length read-back
_text_streams = {}


_fixed_len = threading.local()


@contextlib.contextmanager
def fixed_text_length(length):
    """Inside: the text tower runs on the first `length` positions, a HOST-known bound (>= 1 + the largest EOT position of
    every batch it will see) instead of the per-batch read-back -- what a captured graph needs (graph_step.py). Any
    bound >= the longest caption gives bit-identical rows (causal mask). None = no bound."""
    prev = getattr(_fixed_len, 'value', None)
    _fixed_len.value = None if length is None else int(length)
    try:
        yield
    finally:
        _fixed_len.value = prev


def caption_bound(host_tokens, bucket=8, context=None):
    """1 + the largest EOT position (EOT = the highest id: models.py:158-160) of a HOST token tensor, rounded up to `bucket`
    positions -- the bound fixed_text_length() takes. A driver that has the tokens on the host anyway (the reference's loop
    uploads them itself, main_pretrain.py:486-498) gets the caption trim without the per-step device read-back:

        with models.fixed_text_length(models.caption_bound(inputs[1])):      # BEFORE inputs[1].cuda()
            outputs = model(*inputs_on_device, ...)

    The rounding keeps the number of distinct GEMM shapes (and, for graph_step.py, of captured graphs) small."""
    if host_tokens.is_cuda:
        raise ValueError('caption_bound reads the token tensor on the host; for device tensors the model reads the '
                         'length back itself (one sync per step)')
    longest = int(host_tokens.argmax(dim=-1).max()) + 1
    ctx_len = int(host_tokens.shape[-1]) if context is None else int(context)
    b = max(1, int(bucket))
    return min(ctx_len, (longest + b - 1) // b * b)


_lmax_memos = weakref.WeakKeyDictionary()      # model -> (weakref to the token tensor, its version, longest caption)


def _longest_caption(owner, text, rows, rows_max=None):
    """1 + the largest EOT position of the batch. One scalar read back from the device (`rows_max`: the reduction
    already enqueued by the caller, so that the read-back finds it finished); memoised per model on the identity (weak
    reference) and version of the token tensor, so calling encode_text twice on one batch reads it once. The memo lives
    beside the model (not on it: the model stays picklable), one slot per model instance."""
    memo = _lmax_memos.get(owner)
    if memo is not None and memo[0]() is text and memo[1] == text._version:
        return memo[2]
    lmax = int((rows.max() if rows_max is None else rows_max).item()) + 1
    _lmax_memos[owner] = (weakref.ref(text), text._version, lmax)
    return lmax


def _text_stream(device):
    """One side stream per device for the text tower (kept out of the module so that the model stays picklable).
    (A lower-than-default priority would suit it -- 3 % of the work, off the critical path -- but the priority range of
    this runtime is (0, -1): default IS the lowest; measured identical either way.)"""
    st = _text_streams.get(device)
    if st is None:
        st = _text_streams[device] = torch.cuda.Stream(device=device)
    return st


class CLIP(nn.Module):
    def __init__(self,
                 embed_dim: int,
                 vision_width: int,
                 vision_model: nn.Module,
                 context_length: int,
                 vocab_size: int,
                 transformer_width: int,
                 transformer_heads: int,
                 transformer_layers: int,
                 tempearture_init=0.07,      # [sic] the reference spells it this way (models.py:86)
                 **kwargs):
        super().__init__()
        self.context_length = context_length
        self.vision_width = vision_width
        self.visual = vision_model
        self.transformer = Transformer(width=transformer_width, layers=transformer_layers, heads=transformer_heads,
                                       attn_mask=self.build_attention_mask())
        self.vocab_size = vocab_size
        self.token_embedding = nn.Embedding(vocab_size, transformer_width)
        self.positional_embedding = nn.Parameter(torch.empty(self.context_length, transformer_width))
        self.ln_final = LayerNorm(transformer_width)
        self.image_projection = nn.Parameter(torch.empty(vision_width, embed_dim))
        self.text_projection = nn.Parameter(torch.empty(transformer_width, embed_dim))
        print("=> initialize initial temperature with {}".format(tempearture_init))
        self.logit_scale = nn.Parameter(torch.ones([]) * np.log(1 / tempearture_init))
        self.initialize_parameters()

    def initialize_parameters(self):
        """models.py:115-129"""
        nn.init.normal_(self.token_embedding.weight, std=0.02)
        nn.init.normal_(self.positional_embedding, std=0.01)
        width, layers = self.transformer.width, self.transformer.layers
        proj_std = (width ** -0.5) * ((2 * layers) ** -0.5)
        for block in self.transformer.resblocks:
            nn.init.normal_(block.attn.in_proj_weight, std=width ** -0.5)
            nn.init.normal_(block.attn.out_proj.weight, std=proj_std)
            nn.init.normal_(block.mlp.c_fc.weight, std=(2 * width) ** -0.5)
            nn.init.normal_(block.mlp.c_proj.weight, std=proj_std)
        nn.init.normal_(self.image_projection, std=self.vision_width ** -0.5)
        nn.init.normal_(self.text_projection, std=width ** -0.5)

    def build_attention_mask(self):
        """Additive causal mask (models.py:131-137); the HIP kernel applies it analytically."""
        mask = torch.empty(self.context_length, self.context_length)
        mask.fill_(float("-inf"))
        mask.triu_(1)
        return mask

    def encode_image(self, image, use_checkpoint=False, apply_project=True, _after_block=None):
        """models.py:139-148. `_after_block` (not in the reference signature) is handed to the video tower, see
        SpaceTimeTransformer._features_from_tokens."""
        with _amp_region():
            if _after_block is not None:
                x = self.visual(image, use_checkpoint=use_checkpoint, _after_block=_after_block)
            else:
                x = self.visual(image, use_checkpoint=use_checkpoint)
            if isinstance(x, list):
                assert len(x) == 1
                x = x[0]
            if not apply_project:
                return _half_out(x, image, self.image_projection)
            if x.dtype != self.image_projection.dtype and not torch.is_autocast_enabled():
                x = x.to(self.image_projection.dtype)        # fp16 clip into an f32 model, or the reverse
            return _half_out(ops.project(x, self.image_projection), image, self.image_projection)

    def _eot_rows(self, text):
        """(EOT row of every caption, their maximum as a device scalar or None): enqueued without a host read."""
        rows = text.argmax(dim=-1)
        trim = (_TEXT_TRIM and text.is_cuda and getattr(_fixed_len, 'value', None) is None
                and not torch.cuda.is_current_stream_capturing())
        return rows, (rows.max() if trim else None)

    def encode_text(self, text, use_checkpoint=False, _eot=None):
        with ops.model_forward(), _amp_region():
            # Only the EOT row (highest token id, models.py:158-160) of the last layer feeds the output, and under
            # the causal mask a row never sees later positions: everything behind the longest caption of the batch
            # is dead work in every layer (and receives exactly zero gradient in the reference too). The tower runs
            # on columns [0, max EOT] only -- bit-for-bit the same rows, 32/77 of the work on 32-token captions.
            # Costs one host read of a scalar (the reference driver reads loss.item() every step anyway);
            # LAVILA_TEXT_TRIM=0 (or stream capture) keeps all 77 positions.
            rows, rows_max = self._eot_rows(text) if _eot is None else _eot
            bound = getattr(_fixed_len, 'value', None)
            if bound is not None:
                text = text[:, :max(1, min(bound, text.shape[1]))]
                # a bound below the real longest caption is the caller's bug; it must not become an out-of-range gather
                rows = rows.clamp(max=text.shape[1] - 1)
            elif rows_max is not None:
                text = text[:, :_longest_caption(self, text, rows, rows_max)]
            emb = self.token_embedding
            plain = (type(emb) is nn.Embedding and emb.padding_idx is None and emb.max_norm is None
                     and not emb.scale_grad_by_freq and not emb.sparse and not emb._forward_hooks
                     and not emb._forward_pre_hooks)
            act = (ops.autocast_dtype() if (torch.is_autocast_enabled() and not ops.RESIDUAL_F32) else None) or torch.float32
            x = ops.text_embed(text, emb.weight, self.positional_embedding, act) if plain else None    # gather + add, one kernel
            if x is None:
                x = self.token_embedding(text) + self.positional_embedding[:text.shape[1]]      # [B, L, W]
                if x.dtype == torch.float16:               # model.half(): compute in bf16 (f32 stream if asked for)
                    x = x.float() if ops.RESIDUAL_F32 else x.to(torch.bfloat16)
                elif torch.is_autocast_enabled() and not ops.RESIDUAL_F32:
                    x = x.to(ops.autocast_dtype())
            x = self.transformer.forward_batch_major(x, self.ln_final, use_checkpoint=use_checkpoint, rows=rows)
            if x.dtype != self.text_projection.dtype and not torch.is_autocast_enabled():
                x = x.to(self.text_projection.dtype)
            return _half_out(ops.project(x, self.text_projection), self.text_projection)

    def forward(self, image, text, use_checkpoint=False, norm_embed=False):
        with ops.model_forward():
            return self._forward(image, text, use_checkpoint, norm_embed)

    def _forward(self, image, text, use_checkpoint, norm_embed):
        if _TEXT_STREAM and image.is_cuda:
            # The two towers are independent until the loss: the (small) text tower runs on a second HIP stream so that
            # its short kernels fill the tails of the video tower's launches. Autograd replays each tower's backward on
            # the stream its forward ran on and joins the streams at the end of backward().
            main = torch.cuda.current_stream(image.device)
            side = _text_stream(image.device)
            side.wait_stream(main)
            text.record_stream(side)
            # The text tower needs the longest caption length on the host, and that read-back has to wait for the
            # PREVIOUS step to drain (the token tensor was produced behind it on the main stream). So: enqueue the
            # reduction, enqueue the first blocks of the video tower, and only then read the length and enqueue the text
            # tower (from a hook inside the video forward): while the host waits the GPU still has work queued, and the
            # text tower starts early enough to finish long before the video tower does.
            with torch.cuda.stream(side):
                eot = self._eot_rows(text)
            box = []

            def text_tower():
                with torch.cuda.stream(side):
                    t = self.encode_text(text, use_checkpoint=use_checkpoint, _eot=eot)
                    box.append(F.normalize(t.float(), dim=-1) if norm_embed else t)

            if isinstance(self.visual, SpaceTimeTransformer):
                # the hook travels as an argument of this call (no module state: concurrent forwards are independent)
                image_embed = self.encode_image(image, use_checkpoint=use_checkpoint,
                                                _after_block=(_TEXT_AFTER_BLOCK, text_tower))
            else:
                text_tower()
                image_embed = self.encode_image(image, use_checkpoint=use_checkpoint)
            if not box:                       # fewer blocks than the hook position
                text_tower()
            text_embed = box[0]
            if norm_embed:
                image_embed = F.normalize(image_embed.float(), dim=-1)
            main.wait_stream(side)
            text_embed.record_stream(main)
            return {'image_embed': image_embed, 'text_embed': text_embed, 'logit_scale': self.logit_scale.exp()}
        image_embed = self.encode_image(image, use_checkpoint=use_checkpoint)
        text_embed = self.encode_text(text, use_checkpoint=use_checkpoint)
        if norm_embed:
            image_embed = F.normalize(image_embed.float(), dim=-1)
            text_embed = F.normalize(text_embed.float(), dim=-1)
        return {'image_embed': image_embed,
                'text_embed': text_embed,
                'logit_scale': self.logit_scale.exp()}


def get_loss(model, args, tokenizer=None):
    if model.startswith('CLIP'):
        return loss.CLIPLoss(use_vissl=args.contrastive_use_vissl, cache_labels=True, rank=args.rank,
                             world_size=args.world_size)
    raise NotImplementedError(f'{model}: only the CLIP_* dual-encoder path is built (SURVEY.md section 8)')


def get_metric_names(model):
    if model.startswith('CLIP'):
        return ['loss', 'clip_loss', 'clip_acc']
    raise NotImplementedError(f'{model}: only the CLIP_* dual-encoder path is built (SURVEY.md section 8)')


# ------------------------------------------------------------------------------------------------------
# named constructors
# ------------------------------------------------------------------------------------------------------
def load_openai_clip(name, device='cpu'):
    """The reference downloads OpenAI CLIP weights here (openai_clip.py:104-146). Offline there is nothing to
    download: a local TorchScript/state_dict file can be given through LAVILA_CLIP_WEIGHTS_DIR; otherwise the
    caller keeps its (seeded) initialisation."""
    root = os.environ.get('LAVILA_CLIP_WEIGHTS_DIR')
    if not root:
        import warnings
        warnings.warn(f'lavila_amd: OpenAI CLIP weights for {name} are NOT loaded (no network; set '
                      'LAVILA_CLIP_WEIGHTS_DIR to a directory holding the .pt files): the model keeps its seeded '
                      'random initialisation, unlike the reference constructor', stacklevel=3)
        return None
    path = os.path.join(root, name.replace('/', '-') + '.pt')
    if not os.path.isfile(path):
        raise RuntimeError(f'Model {name} not found at {path}')
    obj = torch.load(path, map_location=device, weights_only=False)
    return obj.state_dict() if hasattr(obj, 'state_dict') else obj


def _load_clip_weights(model, vision_model, clip_sd, layers, project_embed_dim):
    """Copies OpenAI-CLIP weights the way models.py:329-370 does (vision via remap_keys, strict=False)."""
    vis = {k[len('visual.'):]: v for k, v in clip_sd.items() if k.startswith('visual.')}
    print(vision_model.load_state_dict(remap_keys(vis, transformer_layers=layers), strict=False))
    model.transformer.load_state_dict({k[len('transformer.'):]: v for k, v in clip_sd.items()
                                       if k.startswith('transformer.')})
    model.token_embedding.load_state_dict({'weight': clip_sd['token_embedding.weight']})
    model.positional_embedding.data.copy_(clip_sd['positional_embedding'])
    model.ln_final.load_state_dict({'weight': clip_sd['ln_final.weight'], 'bias': clip_sd['ln_final.bias']})
    if project_embed_dim == clip_sd['text_projection'].shape[1]:
        print("=> Loading CLIP's text_projection, image_projection and logit_scale directly")
        model.image_projection.data.copy_(clip_sd['visual.proj'])
        model.text_projection.data.copy_(clip_sd['text_projection'])
        model.logit_scale.data.copy_(clip_sd['logit_scale'])


def _clip_openai_timesformer(clip_name, vision_kwargs, vision_width, text_width, text_heads, vision_layers,
                             num_frames, timesformer_gated_xattn, drop_path_rate, timesformer_freeze_space,
                             temperature_init, project_embed_dim, kwargs):
    vision_model = SpaceTimeTransformer(
        num_frames=num_frames, time_init='zeros', attention_style='frozen-in-time', ln_pre=True,
        act_layer=QuickGELU, is_tanh_gating=timesformer_gated_xattn, drop_path_rate=drop_path_rate,
        **vision_kwargs)
    clip_sd = load_openai_clip(clip_name, 'cpu')
    pretrained_names = set()
    if clip_sd is not None:
        print(f"=> Loading CLIP ({clip_name}) weights")
        vis = {k[len('visual.'):]: v for k, v in clip_sd.items() if k.startswith('visual.')}
        pretrained_names = set(remap_keys(dict(vis), transformer_layers=vision_layers).keys())
    if timesformer_freeze_space and clip_sd is None:
        raise RuntimeError('timesformer_freeze_space=True needs the pretrained CLIP weights (nothing would be frozen '
                           'without them); set LAVILA_CLIP_WEIGHTS_DIR')
    if timesformer_freeze_space:
        print("=> Freeze the space part in TimeSformer")
        freeze_list, unfreeze_list = [], []
        for n, p in vision_model.named_parameters():
            if n not in pretrained_names or n == 'cls_token':
                p.requires_grad = True
                unfreeze_list.append(n)
            else:
                p.requires_grad = False
                freeze_list.append(n)
        print("Freeze the pretrained parts in TimeSformer: {}".format(freeze_list))
        print(" Learn the rest parts in TimeSformer: {}".format(unfreeze_list))
    vision_model.head = nn.Identity()
    vision_model.pre_logits = nn.Identity()
    vision_model.fc = nn.Identity()
    model = CLIP(embed_dim=project_embed_dim, vision_width=vision_width, vision_model=vision_model,
                 context_length=77, vocab_size=49408, transformer_width=text_width,
                 transformer_heads=text_heads, transformer_layers=12, tempearture_init=temperature_init, **kwargs)
    if clip_sd is not None:
        _load_clip_weights(model, vision_model, clip_sd, vision_layers, project_embed_dim)
    return model


def CLIP_OPENAI_TIMESFORMER_BASE(
    num_frames=4, timesformer_gated_xattn=False, drop_path_rate=0, timesformer_freeze_space=False,
    temperature_init=0.07, project_embed_dim=256, **kwargs,
):
    """models.py:316-371: TSF-B/16 @224 (D=768, depth 12, 12 heads) + CLIP text (512 wide, 8 heads, 12 layers)."""
    return _clip_openai_timesformer('ViT-B/16', {}, 768, 512, 8, 12, num_frames, timesformer_gated_xattn,
                                    drop_path_rate, timesformer_freeze_space, temperature_init, project_embed_dim,
                                    kwargs)


def CLIP_OPENAI_TIMESFORMER_LARGE(
    num_frames=4, timesformer_gated_xattn=False, drop_path_rate=0, timesformer_freeze_space=False,
    temperature_init=0.07, project_embed_dim=256, **kwargs,
):
    """models.py:374-431: TSF-L/14 @224 (D=1024, depth 24, 16 heads) + CLIP text (768 wide, 12 heads)."""
    vk = dict(img_size=224, patch_size=14, embed_dim=1024, depth=24, num_heads=16)
    return _clip_openai_timesformer('ViT-L/14', vk, 1024, 768, 12, 24, num_frames, timesformer_gated_xattn,
                                    drop_path_rate, timesformer_freeze_space, temperature_init, project_embed_dim,
                                    kwargs)


def CLIP_OPENAI_TIMESFORMER_LARGE_336PX(
    num_frames=4, timesformer_gated_xattn=False, drop_path_rate=0, timesformer_freeze_space=False,
    temperature_init=0.07, project_embed_dim=256, **kwargs,
):
    """models.py:434-491: TSF-L/14 @336."""
    vk = dict(img_size=336, patch_size=14, embed_dim=1024, depth=24, num_heads=16)
    return _clip_openai_timesformer('ViT-L/14@336px', vk, 1024, 768, 12, 24, num_frames, timesformer_gated_xattn,
                                    drop_path_rate, timesformer_freeze_space, temperature_init, project_embed_dim,
                                    kwargs)


# --------------------------------------------------------------------------------------------------
# narrator constructors (inference): TimeSformer tower + attention pooling + gated-cross-attention GPT-2
# --------------------------------------------------------------------------------------------------
def load_pretrained_gpt2(name):
    """The reference fetches the GPT-2 checkpoint from the hub (`GPT2LMHeadModel.from_pretrained`, models.py:914-917).
    Offline: LAVILA_GPT2_WEIGHTS_DIR/<name>.pt holding the HF state_dict, else the decoder keeps its initialisation."""
    root = os.environ.get('LAVILA_GPT2_WEIGHTS_DIR')
    if not root:
        import warnings
        warnings.warn(f'lavila_amd: pretrained {name} weights are NOT loaded (no network; set LAVILA_GPT2_WEIGHTS_DIR): '
                      'the decoder keeps its random initialisation, unlike the reference constructor', stacklevel=3)
        return None
    path = os.path.join(root, name + '.pt')
    if not os.path.isfile(path):
        raise RuntimeError(f'GPT-2 checkpoint {name} not found at {path}')
    return torch.load(path, map_location='cpu', weights_only=True)


_XATTN_TAGS = ('crossattention', 'ln_cross_attn', 'alpha_cattn', 'alpha_dense')


def load_gpt2_weights(text_decoder, gpt2_sd):
    """models.py:919-923: EVERY parameter of the plain GPT-2 is written into the gated decoder (`rsetattr(text_decoder,
    n + '.data', p.data)` raises on a name it cannot resolve). Accepted key spellings: the LM-head model's
    (`transformer.h.0...`, what `GPT2LMHeadModel.state_dict()` holds) and the bare `GPT2Model`'s of the hub's raw gpt2
    checkpoints (`h.0...`, `wte.weight`). Fails loudly if a parameter of the plain GPT-2 part of the decoder -- everything
    except the cross-attention additions -- is missing from the checkpoint, has another shape, or if the checkpoint holds
    a parameter the decoder does not know."""
    own = dict(text_decoder.named_parameters())
    plain = {n for n in own if not any(t in n for t in _XATTN_TAGS)}
    loaded = set()
    with torch.no_grad():
        for n, v in gpt2_sd.items():
            if n.endswith(('.attn.bias', '.attn.masked_bias')):          # causal-mask buffers of the HF modules
                continue
            key = n if n in own else 'transformer.' + n
            if n == 'lm_head.weight':                                     # tied to transformer.wte.weight
                if tuple(v.shape) != tuple(own['transformer.wte.weight'].shape):
                    raise RuntimeError(f'GPT-2 checkpoint: lm_head.weight {tuple(v.shape)} does not fit the decoder')
                continue
            if key not in own:
                raise RuntimeError(f'GPT-2 checkpoint: parameter {n!r} has no counterpart in the gated decoder')
            if tuple(own[key].shape) != tuple(v.shape):
                raise RuntimeError(f'GPT-2 checkpoint: {n!r} is {tuple(v.shape)}, the decoder expects '
                                   f'{tuple(own[key].shape)}')
            own[key].copy_(v)
            loaded.add(key)
    missing = sorted(plain - loaded)
    if missing:
        raise RuntimeError(f'GPT-2 checkpoint: {len(missing)} parameters of the plain GPT-2 were not found '
                           f'(first: {missing[:3]}); the decoder would keep random weights there')
    return loaded


def _vclm_openai_timesformer(clip_name, vision_kwargs, vision_width, vision_layers, gpt2_name, cross_attn_freq, text_width,
                             heads, gated_xattn, random_init_gpt2, freeze_lm_vclm, freeze_visual_vclm,
                             freeze_visual_vclm_temporal, num_frames, timesformer_gated_xattn, kwargs):
    """The common body of models.py:887-1198's TimeSformer narrators."""
    vision_model = SpaceTimeTransformer(
        num_frames=num_frames, time_init='zeros', attention_style='frozen-in-time', ln_pre=True,
        act_layer=QuickGELU, is_tanh_gating=timesformer_gated_xattn, **vision_kwargs)
    clip_sd = load_openai_clip(clip_name, 'cpu')
    if clip_sd is not None:
        print(f"=> Loading CLIP ({clip_name}) weights")
        vis = {k[len('visual.'):]: v for k, v in clip_sd.items() if k.startswith('visual.')}
        print(vision_model.load_state_dict(remap_keys(vis, transformer_layers=vision_layers), strict=False))
    vision_model.head = nn.Identity()
    vision_model.pre_logits = nn.Identity()
    vision_model.fc = nn.Identity()
    config = augment_gpt2_config(gpt2_config(gpt2_name), cross_attn_freq=cross_attn_freq, gated_xattn=gated_xattn)
    text_decoder = GatedGPT2LMHeadModel(config)
    if not random_init_gpt2:
        gpt2_sd = load_pretrained_gpt2(gpt2_name)
        if gpt2_sd is not None:
            print('Loading LM from pretrained weights..')
            load_gpt2_weights(text_decoder, gpt2_sd)
    if freeze_lm_vclm:
        print('Freeze the LM part of TextDecoder of VCLM')
        text_decoder.freeze_lm_weights()
    if freeze_visual_vclm:
        print('Freeze the spatial part of VideoEncoder of VCLM')
        vision_model.freeze_spatial_weights()
    if freeze_visual_vclm_temporal:
        print('Freeze the temporal part of VideoEncoder of VCLM')
        vision_model.freeze_temporal_weights()
    return VCLM_HF(vision_width=vision_width, vision_model=vision_model, text_width=text_width,
                   text_decoder=text_decoder, num_img_queries=256, dim_head=64, heads=heads, **kwargs)


_TSF_L14 = dict(img_size=224, patch_size=14, embed_dim=1024, depth=24, num_heads=16)
_TSF_L14_336 = dict(img_size=336, patch_size=14, embed_dim=1024, depth=24, num_heads=16)


def VCLM_OPENAI_TIMESFORMER_BASE_GPT2(gated_xattn=False, random_init_gpt2=False, freeze_lm_vclm=False,
                                      freeze_visual_vclm=False, freeze_visual_vclm_temporal=False, num_frames=4,
                                      timesformer_gated_xattn=False, **kwargs):
    """models.py:887-948: TSF-B/16 + GPT-2 (768 wide, 12 layers), cross-attention in every block."""
    return _vclm_openai_timesformer('ViT-B/16', {}, 768, 12, 'gpt2', 1, 768, 12, gated_xattn, random_init_gpt2,
                                    freeze_lm_vclm, freeze_visual_vclm, freeze_visual_vclm_temporal, num_frames,
                                    timesformer_gated_xattn, kwargs)


def VCLM_OPENAI_TIMESFORMER_BASE_GPT2_XL(gated_xattn=False, freeze_lm_vclm=False, freeze_visual_vclm=False,
                                         freeze_visual_vclm_temporal=False, num_frames=4, timesformer_gated_xattn=False,
                                         **kwargs):
    """models.py:951-1009: TSF-B/16 + GPT-2 XL (1600 wide, 48 layers), cross-attention in every 2nd block."""
    return _vclm_openai_timesformer('ViT-B/16', {}, 768, 12, 'gpt2-xl', 2, 1600, 25, gated_xattn, False, freeze_lm_vclm,
                                    freeze_visual_vclm, freeze_visual_vclm_temporal, num_frames, timesformer_gated_xattn,
                                    kwargs)


def VCLM_OPENAI_TIMESFORMER_LARGE_GPT2_XL(gated_xattn=False, freeze_lm_vclm=False, freeze_visual_vclm=False,
                                          freeze_visual_vclm_temporal=False, num_frames=4, timesformer_gated_xattn=False,
                                          **kwargs):
    """models.py:1012-1072: TSF-L/14 + GPT-2 XL, cross-attention in every 2nd block."""
    return _vclm_openai_timesformer('ViT-L/14', dict(_TSF_L14), 1024, 24, 'gpt2-xl', 2, 1600, 25, gated_xattn, False,
                                    freeze_lm_vclm, freeze_visual_vclm, freeze_visual_vclm_temporal, num_frames,
                                    timesformer_gated_xattn, kwargs)


def VCLM_OPENAI_TIMESFORMER_LARGE_GPT2(gated_xattn=False, freeze_lm_vclm=False, freeze_visual_vclm=False,
                                       freeze_visual_vclm_temporal=False, num_frames=4, timesformer_gated_xattn=False,
                                       **kwargs):
    """models.py:1075-1135: TSF-L/14 + GPT-2, cross-attention in every block."""
    return _vclm_openai_timesformer('ViT-L/14', dict(_TSF_L14), 1024, 24, 'gpt2', 1, 768, 12, gated_xattn, False,
                                    freeze_lm_vclm, freeze_visual_vclm, freeze_visual_vclm_temporal, num_frames,
                                    timesformer_gated_xattn, kwargs)


def VCLM_OPENAI_TIMESFORMER_LARGE_336PX_GPT2_XL(gated_xattn=False, freeze_lm_vclm=False, freeze_visual_vclm=False,
                                                freeze_visual_vclm_temporal=False, num_frames=4,
                                                timesformer_gated_xattn=False, **kwargs):
    """models.py:1138-1198: TSF-L/14 @336 + GPT-2 XL, cross-attention in every 3rd block."""
    return _vclm_openai_timesformer('ViT-L/14@336px', dict(_TSF_L14_336), 1024, 24, 'gpt2-xl', 3, 1600, 25, gated_xattn,
                                    False, freeze_lm_vclm, freeze_visual_vclm, freeze_visual_vclm_temporal, num_frames,
                                    timesformer_gated_xattn, kwargs)
