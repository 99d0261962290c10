"""Checkpoint-format helpers with the reference's names (lavila/models/utils.py).

These are host-side state_dict transforms (no device work): the temporal-embedding inflation used by
eval_zeroshot.py:89-97 and the CLIP-ViT -> TimeSformer key mapping used by the named constructors
(models.py:331-332). They define the on-disk naming contract.
"""
from collections import OrderedDict
import functools

import torch
import torch.nn.functional as F


def inflate_positional_embeds(current_model_state_dict, new_state_dict, num_frames=4, load_temporal_fix='bilinear'):
    """lavila/models/utils.py:13-59: adapt 'visual.temporal_embed' [1,F_ckpt,D] to this model's frame count
    (truncate, zero-extend, or nearest/bilinear interpolate); spatial size changes are rejected."""
    te = 'visual.temporal_embed'
    if te in new_state_dict and te in current_model_state_dict:
        loaded = new_state_dict[te]
        have, want, dim = loaded.shape[1], num_frames, loaded.shape[2]
        if have > want:
            print(f'### loaded SpaceTimeTransformer model has MORE frames than current...'
                  f'### loading weights, filling in the extras via {load_temporal_fix}')
            new_state_dict[te] = loaded[:, :want, :]
        elif have < want:
            print(f'### loaded SpaceTimeTransformer model has FEWER frames than current...'
                  f'### loading weights, filling in the extras via {load_temporal_fix}')
            if load_temporal_fix == 'zeros':
                grown = torch.zeros([loaded.shape[0], want, dim])
                grown[:, :have] = loaded
            elif load_temporal_fix in ('interp', 'bilinear'):
                mode = 'bilinear' if load_temporal_fix == 'bilinear' else 'nearest'
                grown = F.interpolate(loaded.unsqueeze(0), (want, dim), mode=mode).squeeze(0)
            else:
                raise NotImplementedError
            new_state_dict[te] = grown
    pe = 'visual.pos_embed'
    if pe in new_state_dict and pe in current_model_state_dict:
        if new_state_dict[pe].shape[1] != current_model_state_dict[pe].shape[1]:
            raise NotImplementedError(
                'Loading models with different spatial resolution / patch number not yet implemented, sorry.')
    return new_state_dict


def rgetattr(obj, attr, *args):
    return functools.reduce(lambda o, a: getattr(o, a, *args), [obj] + attr.split('.'))


def rsetattr(obj, attr, val):
    pre, _, post = attr.rpartition('.')
    return setattr(rgetattr(obj, pre) if pre else obj, post, val)


_VIT_TOP = {
    'class_embedding': 'cls_token', 'positional_embedding': 'pos_embed', 'conv1.weight': 'patch_embed.proj.weight',
    'ln_pre.weight': 'ln_pre.weight', 'ln_pre.bias': 'ln_pre.bias', 'ln_post.weight': 'norm.weight',
    'ln_post.bias': 'norm.bias',
}
_VIT_BLOCK = {
    'attn.in_proj_weight': 'attn.qkv.weight', 'attn.in_proj_bias': 'attn.qkv.bias',
    'attn.out_proj.weight': 'attn.proj.weight', 'attn.out_proj.bias': 'attn.proj.bias',
    'ln_1.weight': 'norm1.weight', 'ln_1.bias': 'norm1.bias', 'ln_2.weight': 'norm2.weight',
    'ln_2.bias': 'norm2.bias', 'mlp.c_fc.weight': 'mlp.fc1.weight', 'mlp.c_fc.bias': 'mlp.fc1.bias',
    'mlp.c_proj.weight': 'mlp.fc2.weight', 'mlp.c_proj.bias': 'mlp.fc2.bias',
}


def remap_keys(clip_state_dict, transformer_layers=12):
    """lavila/models/utils.py:74-108: OpenAI CLIP ViT keys -> SpaceTimeTransformer keys; 'proj' is skipped,
    class/positional embeddings gain their leading singleton dims."""
    out = OrderedDict()
    for key, val in clip_state_dict.items():
        if key == 'proj':
            continue
        if key in _VIT_TOP:
            if key == 'class_embedding':
                val = val.unsqueeze(0).unsqueeze(0)
            elif key == 'positional_embedding':
                val = val.unsqueeze(0)
            out[_VIT_TOP[key]] = val
            continue
        prefix = 'transformer.resblocks.'
        if not key.startswith(prefix):
            raise KeyError(key)
        layer, _, rest = key[len(prefix):].partition('.')
        if int(layer) >= transformer_layers or rest not in _VIT_BLOCK:
            raise KeyError(key)
        out[f'blocks.{layer}.{_VIT_BLOCK[rest]}'] = val
    return out
