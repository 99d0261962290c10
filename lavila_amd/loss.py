"""Contrastive (InfoNCE) loss of the dual encoder behind the reference interface
(`lavila/models/loss.py`: gather_features :18-43, CLIPLoss :46-118).

Same constructor, same `criterion(outputs) -> {'loss','clip_loss','clip_acc'}` contract, same numbers.
The plan is MI355X-first instead of a translation of the reference's NCCL call pattern:

  reference (vissl path)                         here
  ------------------------------------------    -----------------------------------------------------
  2 x all_gather (img, txt), list + cat          1 x all_gather_into_tensor of [B, 2E] (img | txt)
  every rank computes the full G x G logits      each rank computes its two [B, G] slabs (its rows of
  and both cross-entropies (O(G^2) per rank)     logits_per_image and of logits_per_text): O(G^2 / W)
  backward: 2 x all_reduce of [W,B,E] grads      backward: none -- one tiny all_gather of the 2B row
  (+ slice own rank)                             LSEs (+3 partial sums) in forward makes the local
                                                 gradient computable from the gathered embeddings alone

Gradient convention (pinned by tests/golden/clip_loss_multirank.pt): with
`use_vissl=True` the reference hands each rank W x d(global loss)/d(local embeddings) (GatherLayer
backward sums W identical copies; DDP's 1/W averaging restores the true gradient); with the default
`gather_features` path it hands 1 x. Both are reproduced. d(loss)/d(logit_scale) is the full global
derivative on every rank in both modes, as in the reference.
"""
import torch
import torch.distributed as dist
import torch.nn as nn

from . import ops
from .distributed_utils import all_gather_rows


class _ContrastiveFn(torch.autograd.Function):

    @staticmethod
    def forward(ctx, image_embed, text_embed, logit_scale, crit):
        W, rank = crit.world_size, crit.rank
        B, E = image_embed.shape
        dt = image_embed.dtype if (image_embed.dtype == text_embed.dtype and
                                   image_embed.dtype in (torch.float32, torch.bfloat16)) else torch.float32
        img, txt = image_embed.detach().to(dt), text_embed.detach().to(dt)
        scale = logit_scale.detach().float().reshape(1).contiguous()
        if W > 1:
            both = all_gather_rows(torch.cat([img, txt], dim=1))           # [G, 2E], rank-ordered
            img_all, txt_all = both[:, :E].contiguous(), both[:, E:].contiguous()
        else:
            img_all, txt_all = img.contiguous(), txt.contiguous()
        G = img_all.shape[0]
        row0 = rank * B if W > 1 else 0

        stats, argmax = crit._slab_forward(img_all, txt_all, scale, B, row0)   # [2,B,4] f32, [2,B] i32
        lse, diag, expect = stats[..., 0], stats[..., 1], stats[..., 2]
        labels = torch.arange(row0, row0 + B, device=argmax.device, dtype=torch.int32)
        part = torch.stack([(lse - diag).sum(), (expect - diag).sum(),
                            (argmax[0] == labels).sum().float()])
        # local_loss (loss.py:86-88, 99-100; only on the gather_features path): every rank's loss is the mean over
        # ITS rows of the two slabs -- exactly what _slab_forward produced -- and is not averaged over ranks
        local = bool(crit.local_loss) and W > 1 and not crit.use_vissl
        if W > 1 and (not local or crit.gather_with_grad):
            packed = torch.cat([lse.reshape(-1), part])[None]                  # [1, 2B+3]
            allp = all_gather_rows(packed)                                     # [W, 2B+3]
            lse_all = allp[:, :2 * B].reshape(W, 2, B).permute(1, 0, 2).reshape(2, G).contiguous()
            sums = allp[:, 2 * B:].sum(0)
        elif W > 1:                            # local rows only: the kernel reads nothing but the own entries
            lse_all = lse.new_zeros(2, G)
            lse_all[:, row0:row0 + B] = lse
            sums = part
        else:
            lse_all, sums = lse.contiguous(), part
        if local:
            sums = part
            loss = part[0] / (2 * B)
            acc = 100.0 * part[2] / B
        else:
            loss = sums[0] / (2 * G)
            acc = 100.0 * sums[2] / G
        ctx.save_for_backward(img_all, txt_all, lse_all, scale, sums)
        ctx.cfg = (B, G, row0, W, crit, image_embed.dtype, text_embed.dtype, logit_scale.dtype)
        ctx.local = local
        ctx.mark_non_differentiable(acc)
        crit._last_pred = argmax[0]
        return loss, acc

    @staticmethod
    def backward(ctx, dloss, dacc):
        img_all, txt_all, lse_all, scale, sums = ctx.saved_tensors
        B, G, row0, W, crit, idt, tdt, sdt = ctx.cfg
        up = dloss.detach().float().reshape(1).contiguous()
        if ctx.local:
            # with gather_with_grad the all-gather's backward sums, on the owner of a row, the column terms of EVERY
            # rank's local loss (each weighted 1/(2B)): rows + columns over the global batch, as in the vissl path;
            # without it the gathered rows are constants and only the row terms of the own loss remain.
            # ASSUMPTION (holds for loss.backward() and any rank-uniform loss scaling such as GradScaler): every rank
            # feeds the same upstream gradient `dloss` -- this rank's value stands in for the other ranks' in the column
            # terms, which is what lets the backward run without a collective. Per-rank loss weights would need the
            # reference's torch.distributed.nn.all_gather backward instead.
            dimg, dtxt = crit._slab_backward(img_all, txt_all, lse_all, scale, up, 1.0 / (2 * B), B, row0,
                                             rows_only=not crit.gather_with_grad)
            dscale = up[0] * sums[1] / (2 * B) / scale[0]
            return dimg.to(idt), dtxt.to(tdt), dscale.to(sdt), None
        mult = float(W) if (crit.use_vissl or crit.gather_with_grad) else 1.0
        dimg, dtxt = crit._slab_backward(img_all, txt_all, lse_all, scale, up, mult / (2 * G), B, row0)
        dscale = up[0] * sums[1] / (2 * G) / scale[0]
        return dimg.to(idt), dtxt.to(tdt), dscale.to(sdt), None


class CLIPLoss(nn.Module):
    """Symmetric InfoNCE over the global batch -- loss.py:46-118."""

    def __init__(self, use_vissl=False, local_loss=False, gather_with_grad=False, cache_labels=False, rank=0,
                 world_size=1):
        super().__init__()
        self.use_vissl = use_vissl
        self.local_loss = local_loss
        self.gather_with_grad = gather_with_grad
        self.cache_labels = cache_labels
        self.rank = rank
        self.world_size = world_size
        # cache state (kept for interface parity; labels are implicit in the slab kernel: row0 + i)
        self.prev_num_logits = 0
        self.labels = {}
        self._last_pred = None

    # -- kernel hooks (tests override these two with the CPU oracle to exercise the collectives on gloo) ---
    def _slab_forward(self, img_all, txt_all, scale, B, row0):
        stats, argmax, _ = ops.clip_loss_fwd_raw(img_all, txt_all, scale, B, row0, want_logits=False)
        return stats, argmax

    def _slab_backward(self, img_all, txt_all, lse_all, scale, upstream, coef, B, row0, rows_only=False):
        return ops.clip_loss_bwd_raw(img_all, txt_all, lse_all, scale, upstream, coef, B, row0, rows_only)

    def forward(self, outputs):
        image_features = outputs['image_embed']
        text_features = outputs['text_embed']
        logit_scale = outputs['logit_scale']
        if self.world_size > 1 and not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError('CLIPLoss(world_size>1) needs an initialised torch.distributed process group')
        if self.world_size > 1 and self.use_vissl and self.local_loss:
            # the reference builds GLOBAL logits on the vissl path and then offsets the labels by num_logits * rank
            # (loss.py:74-79,99-100): out-of-range targets, an error there as well. Not silently turned into the
            # global loss here.
            raise NotImplementedError('CLIPLoss(use_vissl=True, local_loss=True) on several ranks indexes labels out of '
                                      'range in the reference (loss.py:99-100); not reproduced')
        loss, acc = _ContrastiveFn.apply(image_features, text_features, logit_scale, self)
        return {'loss': loss, 'clip_loss': loss, 'clip_acc': acc}

    @torch.no_grad()
    def debug_slabs(self, outputs):
        """Parity-test helper: this rank's scaled logits slabs [2,B,G] (f32), row predictions and labels
        (int64, global column indices) -- what loss.py:78-79,96-105,113 materialise."""
        img, txt = outputs['image_embed'].detach(), outputs['text_embed'].detach()
        B, E = img.shape
        dt = img.dtype if img.dtype in (torch.float32, torch.bfloat16) and img.dtype == txt.dtype else torch.float32
        img, txt = img.to(dt), txt.to(dt)
        scale = outputs['logit_scale'].detach().float().reshape(1).contiguous()
        if self.world_size > 1:
            both = all_gather_rows(torch.cat([img, txt], dim=1))
            img_all, txt_all = both[:, :E].contiguous(), both[:, E:].contiguous()
            row0 = self.rank * B
        else:
            img_all, txt_all, row0 = img.contiguous(), txt.contiguous(), 0
        _, argmax, logits = ops.clip_loss_fwd_raw(img_all, txt_all, scale, B, row0, want_logits=True)
        labels = torch.arange(row0, row0 + B, device=img.device, dtype=torch.long)
        return {'logits': logits, 'pred': argmax.long(), 'labels': labels}


def gather_features(image_features, text_features, local_loss=False, gather_with_grad=False, rank=0, world_size=1):
    """loss.py:18-43 for API completeness (the loss above does not need it)."""
    from .distributed_utils import GatherLayer
    if gather_with_grad:
        return GatherLayer.apply(image_features), GatherLayer.apply(text_features)
    B = image_features.shape[0]
    all_img, all_txt = all_gather_rows(image_features.detach()), all_gather_rows(text_features.detach())
    if not local_loss:
        all_img = torch.cat([all_img[:rank * B], image_features, all_img[(rank + 1) * B:]], 0)
        all_txt = torch.cat([all_txt[:rank * B], text_features, all_txt[(rank + 1) * B:]], 0)
    return all_img, all_txt


class _SSLContrastiveFn(torch.autograd.Function):
    """Slab-parallel SSLCLIPLoss: same exchange plan as _ContrastiveFn (one fused all-gather of [img|txt|ind], one
    small all-gather of row LSEs + partial sums, no backward collective)."""

    @staticmethod
    def forward(ctx, image_embed, text_embed, logit_scale, scale_pseudo, indicators, crit):
        W, rank = crit.world_size, crit.rank
        B, E = image_embed.shape
        dev = image_embed.device
        dt = image_embed.dtype if (image_embed.dtype == text_embed.dtype and
                                   image_embed.dtype in (torch.float32, torch.bfloat16)) else torch.float32
        img, txt = image_embed.detach().to(dt), text_embed.detach().to(dt)
        ind = indicators.detach().to(dev).reshape(-1)
        real, pseudo = logit_scale.detach().float().reshape(()), scale_pseudo.detach().float().reshape(())
        scales3 = torch.stack([pseudo, torch.sqrt(pseudo * real), real]).contiguous()
        if W > 1:
            both = all_gather_rows(torch.cat([img.float(), txt.float(), ind.float()[:, None]], dim=1))
            img_all, txt_all = both[:, :E].to(dt).contiguous(), both[:, E:2 * E].to(dt).contiguous()
            ind_all = both[:, 2 * E].round().to(torch.int32).contiguous()
        else:
            img_all, txt_all, ind_all = img.contiguous(), txt.contiguous(), ind.to(torch.int32).contiguous()
        G = img_all.shape[0]
        row0 = rank * B if W > 1 else 0

        stats, argmax = crit._slab_forward(img_all, txt_all, ind_all, scales3, B, row0)     # [2,B,8], [2,B]
        lse, diag_l, diag_z = stats[..., 0], stats[..., 1], stats[..., 5]
        ind_loc = ind_all[row0:row0 + B]
        labels = torch.arange(row0, row0 + B, device=dev, dtype=torch.int32)
        ok = (argmax[0] == labels)
        bucket_diag = 2 * ind_loc                                                            # mask of (i,i)
        ds = [(stats[..., 2 + k] - diag_z * (bucket_diag == k).float()[None]).sum() for k in range(3)]
        part = torch.stack([(lse - diag_l).sum(), ds[0], ds[1], ds[2], ok.sum().float(),
                            (ok & (ind_loc == 1)).sum().float(), (ok & (ind_loc == 0)).sum().float()])
        if W > 1:
            allp = all_gather_rows(torch.cat([lse.reshape(-1), part])[None])               # [W, 2B+7]
            lse_all = allp[:, :2 * B].reshape(W, 2, B).permute(1, 0, 2).reshape(2, G).contiguous()
            sums = allp[:, 2 * B:].sum(0)
        else:
            lse_all, sums = lse.contiguous(), part
        num_gt = (ind_all == 1).sum().float()
        num_pseudo = (ind_all == 0).sum().float()
        loss = sums[0] / (2 * G)
        acc = 100.0 * sums[4] / G
        acc_gt = 100.0 * sums[5] / num_gt
        acc_pseudo = 100.0 * sums[6] / num_pseudo
        ctx.save_for_backward(img_all, txt_all, ind_all, lse_all, scales3, sums)
        ctx.cfg = (B, G, row0, W, crit, image_embed.dtype, text_embed.dtype, logit_scale.dtype, scale_pseudo.dtype)
        ctx.mark_non_differentiable(acc, acc_gt, acc_pseudo, num_gt, num_pseudo)
        return loss, acc, acc_gt, acc_pseudo, num_gt, num_pseudo

    @staticmethod
    def backward(ctx, dloss, *unused):
        img_all, txt_all, ind_all, lse_all, scales3, sums = ctx.saved_tensors
        B, G, row0, W, crit, idt, tdt, sdt, pdt = ctx.cfg
        mult = float(W) if crit.use_vissl else 1.0
        up = dloss.detach().float().reshape(1).contiguous()
        dimg, dtxt = crit._slab_backward(img_all, txt_all, ind_all, lse_all, scales3, up, mult / (2 * G), B, row0)
        pseudo, geo, real = scales3[0], scales3[1], scales3[2]
        d0, d1, d2 = (up[0] * sums[1 + k] / (2 * G) for k in range(3))     # d loss / d scales3[k]
        dreal = d2 + d1 * 0.5 * geo / real                                   # d sqrt(p r)/dr = sqrt(p r) / (2 r)
        dpseudo = d0 + d1 * 0.5 * geo / pseudo
        return dimg.to(idt), dtxt.to(tdt), dreal.to(sdt), dpseudo.to(pdt), None, None


class SSLCLIPLoss(nn.Module):
    """InfoNCE with per-pair temperature for pseudo-labelled narrations -- loss.py:121-217. `forward(outputs,
    gt_indicators)`; same constructor, same output dict (loss, clip_loss, num_gt, num_pseudo, clip_acc, clip_acc_gt,
    clip_acc_pseudo) and the learnable `logit_scale_pseudo` parameter."""

    def __init__(self, use_vissl=False, local_loss=False, gather_with_grad=False, cache_labels=False, rank=0,
                 world_size=1, scale_init=0.08, freeze_scale=False):
        super().__init__()
        import numpy as np
        self.use_vissl = use_vissl
        self.local_loss = local_loss          # ignored on one rank, as in the reference; see forward() for W > 1
        self.gather_with_grad = gather_with_grad
        self.cache_labels = cache_labels
        self.rank = rank
        self.world_size = world_size
        self.logit_scale_pseudo = nn.Parameter(torch.ones([]) * np.log(1 / scale_init))
        if freeze_scale:
            self.logit_scale_pseudo.requires_grad = False
        self.prev_num_logits = 0
        self.labels = {}

    # -- kernel hooks (tests override these two with the CPU oracle to exercise the collectives on gloo) ---
    def _slab_forward(self, img_all, txt_all, ind_all, scales3, B, row0):
        stats, argmax, _ = ops.ssl_clip_loss_fwd_raw(img_all, txt_all, ind_all, scales3, B, row0)
        return stats, argmax

    def _slab_backward(self, img_all, txt_all, ind_all, lse_all, scales3, upstream, coef, B, row0):
        return ops.ssl_clip_loss_bwd_raw(img_all, txt_all, ind_all, lse_all, scales3, upstream, coef, B, row0)

    def forward(self, outputs, gt_indicators):
        if self.world_size > 1:
            if not self.use_vissl:
                raise NotImplementedError      # as the reference (loss.py:167-168)
            if self.local_loss:
                # the reference offsets the labels by num_logits * rank although its vissl logits are already global
                # (loss.py:185-186): out-of-range targets, an error there too
                raise NotImplementedError('SSLCLIPLoss(local_loss=True) on several ranks indexes labels out of range '
                                          'in the reference (loss.py:185-186); not reproduced')
            if not (dist.is_available() and dist.is_initialized()):
                raise RuntimeError('SSLCLIPLoss(world_size>1) needs an initialised torch.distributed process group')
        loss, acc, acc_gt, acc_pseudo, num_gt, num_pseudo = _SSLContrastiveFn.apply(
            outputs['image_embed'], outputs['text_embed'], outputs['logit_scale'], self.logit_scale_pseudo.exp(),
            gt_indicators, self)
        return {'loss': loss, 'clip_loss': loss, 'num_gt': num_gt.reshape(1).long().cpu(),
                'num_pseudo': num_pseudo.reshape(1).long().cpu(), 'clip_acc': acc, 'clip_acc_gt': acc_gt,
                'clip_acc_pseudo': acc_pseudo}
