"""Cross-rank exchange of local embeddings (mirrors lavila/models/distributed_utils.py:51-89).

One process per GPU; backend 'nccl' is RCCL on ROCm (xGMI inside a node), 'gloo' in the CPU tests.
`gather_from_all` keeps the reference semantics (rank-ordered concatenation, gradients not cut, 0-dim
tensors unsqueezed, identity when not distributed) but is built MI355X-first:
  forward  = one all_gather_into_tensor into a preallocated [W*B, ...] buffer (no list of W tensors, no cat);
  backward = reduce_scatter_tensor(SUM) -- each rank receives only its own slice (1/W of the traffic of the
             reference's all_reduce-then-slice, distributed_utils.py:64-67), same values.
The contrastive loss itself (lavila_amd/loss.py) does not need the backward collective at all.
"""
import torch
import torch.distributed as dist


def is_distributed_training_run() -> bool:
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def all_gather_rows(t: torch.Tensor, group=None) -> torch.Tensor:
    """Non-differentiable rank-ordered gather along dim 0 into one contiguous buffer. Under a GraphedTrainStep capture
    (graph_step._Segments) the collective is not captured: the running graph segment ends in front of it, the gather runs
    eagerly -- now, and between the two replays at every later step, on the same fixed buffers -- and the next segment
    starts behind it."""
    world = dist.get_world_size(group)
    out = t.new_empty((world * t.shape[0],) + tuple(t.shape[1:]))
    src = t.contiguous()
    if t.is_cuda:
        from .graph_step import active_segments, run_on_collective_stream
        op = lambda: run_on_collective_stream(lambda: _all_gather(out, src, group))      # noqa: E731
        if torch.cuda.is_current_stream_capturing():
            seg = active_segments()
            if seg is None:
                raise RuntimeError('all_gather_rows inside a hipGraph capture that lavila_amd.graph_step does not own: '
                                   'collectives are kept between graph segments (GraphedTrainStep), not captured')
            seg.eager(op)
        else:
            op()          # inside a GraphedTrainStep iteration: on its communication stream; otherwise the current stream
        return out
    _all_gather(out, src, group)
    return out


def _all_gather(out, src, group):
    dist.all_gather_into_tensor(out, src, group=group)


class GatherLayer(torch.autograd.Function):
    """all-gather that does not cut gradients (distributed_utils.py:51-67)."""

    @staticmethod
    def forward(ctx, x):
        ctx.rows = x.shape[0]
        return all_gather_rows(x)

    @staticmethod
    def backward(ctx, grad):
        grad = grad.contiguous()
        own = grad.new_empty((ctx.rows,) + tuple(grad.shape[1:]))
        if dist.get_backend() == 'gloo':      # gloo has no reduce_scatter: same result via all_reduce + slice
            dist.all_reduce(grad)
            r = dist.get_rank()
            own.copy_(grad[r * ctx.rows:(r + 1) * ctx.rows])
        else:
            dist.reduce_scatter_tensor(own, grad, op=dist.ReduceOp.SUM)
        return own


def gather_from_all(tensor: torch.Tensor) -> torch.Tensor:
    if tensor.ndim == 0:
        tensor = tensor.unsqueeze(0)
    if is_distributed_training_run():
        return GatherLayer.apply(tensor)
    return tensor
