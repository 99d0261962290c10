"""MI355X-native CLIP text Transformer behind the reference interface
(`lavila/models/openai_model.py`: QuickGELU :177-179, ResidualAttentionBlock :182-216, Transformer :219-232).

Same class names, constructor signatures and state_dict keys (attn.in_proj_weight, attn.in_proj_bias,
attn.out_proj.*, ln_1.*, ln_2.*, mlp.c_fc.*, mlp.c_proj.*). The execution is batch-major ([B, L, W]; the
reference's NLD<->LND permutes disappear), residual adds are fused into the LayerNorms, bias+QuickGELU and its
backward are epilogues of the c_fc / c_proj GEMMs (lvl_linear_tn), and the causal attention core is a C-ABI call
(lvl_causal_attn_fwd/_bwd).
"""
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.utils.checkpoint as checkpoint

from . import ops
from .timesformer import LayerNorm


class QuickGELU(nn.Module):
    """x * sigmoid(1.702 x) -- openai_model.py:177-179 (HIP kernel)."""

    def forward(self, x: torch.Tensor):
        return ops.bias_quick_gelu(x, None)


class _SelfAttentionParams(nn.Module):
    """Parameter container with nn.MultiheadAttention's names (in_proj_weight/in_proj_bias/out_proj)."""

    def __init__(self, d_model: int, n_head: int):
        super().__init__()
        self.embed_dim = d_model
        self.num_heads = n_head
        self.in_proj_weight = nn.Parameter(torch.empty(3 * d_model, d_model))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * d_model))
        self.out_proj = nn.Linear(d_model, d_model)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.constant_(self.out_proj.bias, 0.)


class ResidualAttentionBlock(nn.Module):
    def __init__(self, d_model: int, n_head: int, attn_mask: torch.Tensor = None):
        super().__init__()
        if d_model // n_head != 64:
            raise NotImplementedError('lavila_amd attention kernels are built for head_dim 64')
        self.attn = _SelfAttentionParams(d_model, n_head)
        self.ln_1 = LayerNorm(d_model)
        self.mlp = nn.Sequential(OrderedDict([
            ("c_fc", nn.Linear(d_model, d_model * 4)),
            ("gelu", QuickGELU()),
            ("c_proj", nn.Linear(d_model * 4, d_model))
        ]))
        self.ln_2 = LayerNorm(d_model)
        # The reference passes CLIP.build_attention_mask() (causal, models.py:131-137); the kernel applies the
        # causal mask analytically. Any other mask is rejected loudly.
        if attn_mask is not None:
            L = attn_mask.shape[0]
            causal = torch.full((L, L), float('-inf')).triu_(1)
            if attn_mask.shape != causal.shape or not torch.equal(attn_mask.float().cpu(), causal):
                raise NotImplementedError('only the causal CLIP text mask is supported')
        else:
            raise NotImplementedError('unmasked text attention is not on the LaViLa hot path')
        self.attn_mask = attn_mask

    def chain(self, res, pend, pend_bias):
        """Batch-major block on the fused residual chain (see SpaceTimeBlock.chain)."""
        l1, l2, at = self.ln_1, self.ln_2, self.attn
        if pend is None:
            x = res
            h = ops.layer_norm(x, l1.weight, l1.bias, l1.eps)
        else:
            x, h = ops.add_layer_norm(res, pend, pend_bias, l1.weight, l1.bias, l1.eps, keep_sum=True)
        o = ops.causal_attention(ops.linear(h, at.in_proj_weight, at.in_proj_bias.detach()), at.num_heads,
                                 bias=at.in_proj_bias)
        # x + attn(ln_1(x)) (openai_model.py:199): the residual add rides in out_proj's GEMM epilogue when the shapes
        # are the MFMA GEMM's (ops.RESIDUAL_EPILOGUE), ln_2 reads the sum
        fused = ops.linear_residual_layer_norm(o, at.out_proj.weight, at.out_proj.bias, x, l2.weight, l2.bias, l2.eps)
        if fused is not None:
            x1, h2 = fused
        else:
            y = ops.linear(o, at.out_proj.weight)
            x1, h2 = ops.add_layer_norm(x, y, at.out_proj.bias, l2.weight, l2.bias, l2.eps, keep_sum=True)
        return x1, ops.mlp_quickgelu(h2, self.mlp.c_fc.weight, self.mlp.c_fc.bias, self.mlp.c_proj.weight), \
            self.mlp.c_proj.bias

    def chain_rows(self, res, pend, pend_bias, rows):
        """The LAST block when only one position per caption is read afterwards (the EOT row, models.py:158-160): the
        attention runs on every position (row `rows[b]` attends to all positions before it), the output projection, ln_2
        and the MLP on the B selected rows only -- exact, see SpaceTimeBlock.chain_cls. Returns [B, W] tensors."""
        l1, l2, at = self.ln_1, self.ln_2, self.attn
        if pend is None:
            x = res
            h = ops.layer_norm(x, l1.weight, l1.bias, l1.eps)
        else:
            x, h = ops.add_layer_norm(res, pend, pend_bias, l1.weight, l1.bias, l1.eps, keep_sum=True)
        o = ops.causal_attention(ops.linear(h, at.in_proj_weight, at.in_proj_bias.detach()), at.num_heads,
                                 bias=at.in_proj_bias)
        idx = torch.arange(x.shape[0], device=x.device)
        y = ops.linear(o[idx, rows].contiguous(), at.out_proj.weight)
        x1, h2 = ops.add_layer_norm(x[idx, rows].contiguous(), y, at.out_proj.bias, l2.weight, l2.bias, l2.eps,
                                    keep_sum=True)
        return x1, ops.mlp_quickgelu(h2, self.mlp.c_fc.weight, self.mlp.c_fc.bias, self.mlp.c_proj.weight), \
            self.mlp.c_proj.bias

    def forward(self, x: torch.Tensor, use_checkpoint=False):
        """Reference signature: x is [L, N, D] (openai_model.py:206-216)."""
        xb = x.permute(1, 0, 2).contiguous()
        x1, y, b = self.chain(xb, None, None)
        return (x1 + y + b.to(y.dtype)).permute(1, 0, 2)


class Transformer(nn.Module):
    def __init__(self, width: int, layers: int, heads: int, attn_mask: torch.Tensor = None):
        super().__init__()
        self.width = width
        self.layers = layers
        self.resblocks = nn.Sequential(*[ResidualAttentionBlock(width, heads, attn_mask) for _ in range(layers)])

    def forward_batch_major(self, x, final_ln, use_checkpoint=False, rows=None):
        """x: [B, L, W]. Runs all blocks on the fused chain and applies `final_ln` (CLIP.ln_final). If `rows`
        ([B] int64) is given only those token rows are normalised and returned ([B, W])."""
        from .timesformer import CLS_ONLY_LAST_BLOCK
        res, pend, pb = x.contiguous(), None, None
        last = len(self.resblocks) - 1
        for i, blk in enumerate(self.resblocks):
            if rows is not None and i == last and CLS_ONLY_LAST_BLOCK:      # only the selected rows leave the tower
                if use_checkpoint:
                    res, pend, pb = checkpoint.checkpoint(blk.chain_rows, res, pend, pb, rows, use_reentrant=False)
                else:
                    res, pend, pb = blk.chain_rows(res, pend, pb, rows)
            elif use_checkpoint:
                res, pend, pb = checkpoint.checkpoint(blk.chain, res, pend, pb, use_reentrant=False)
            else:
                res, pend, pb = blk.chain(res, pend, pb)
        if rows is not None:
            idx = torch.arange(res.shape[0], device=res.device)
            r = res if res.dim() == 2 else res[idx, rows].contiguous()
            if pend is None:
                return ops.layer_norm(r, final_ln.weight, final_ln.bias, final_ln.eps)
            p = pend if pend.dim() == 2 else pend[idx, rows].contiguous()
            return ops.add_layer_norm(r, p, pb, final_ln.weight, final_ln.bias, final_ln.eps, keep_sum=False)[1]
        if pend is None:
            return ops.layer_norm(res, final_ln.weight, final_ln.bias, final_ln.eps)
        return ops.add_layer_norm(res, pend, pb, final_ln.weight, final_ln.bias, final_ln.eps, keep_sum=False)[1]

    def forward(self, x: torch.Tensor, use_checkpoint=False):
        """Reference signature: [L, N, D] -> [L, N, D] (openai_model.py:226-232)."""
        res, pend, pb = x.permute(1, 0, 2).contiguous(), None, None
        for blk in self.resblocks:
            res, pend, pb = blk.chain(res, pend, pb)
        if pend is not None:
            res = res + pend + pb.to(pend.dtype)
        return res.permute(1, 0, 2)
