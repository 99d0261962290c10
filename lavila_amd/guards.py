"""Debugging guards of the HIP path (used by the parity tests and by __graft_entry__.smoke())."""
import contextlib

import torch


@contextlib.contextmanager
def forbid_library_gemm():
    """Any GEMM-shaped torch entry point raises: what runs inside the block runs on lavila_amd's own kernels."""
    import torch.nn.functional as F
    names = [(F, 'linear'), (F, 'bilinear'), (F, 'scaled_dot_product_attention'), (F, 'multi_head_attention_forward'),
             (F, 'conv2d'), (F, 'conv3d'),
             (torch, 'matmul'), (torch, 'mm'), (torch, 'bmm'), (torch, 'addmm'), (torch, 'baddbmm'), (torch, 'einsum'),
             (torch, 'mv'), (torch, 'addmv'), (torch, 'tensordot'), (torch, 'inner'), (torch, 'conv2d'),
             (torch.Tensor, 'matmul'), (torch.Tensor, '__matmul__'), (torch.Tensor, '__rmatmul__'), (torch.Tensor, 'mm'),
             (torch.Tensor, 'bmm'), (torch.Tensor, 'addmm'), (torch.Tensor, 'mv')]
    saved = [(o, n, getattr(o, n)) for o, n in names]

    def make(n):
        def boom(*a, **k):
            raise AssertionError(f'library GEMM entry point torch...{n} called on the lavila_amd path')
        return boom
    for o, n, _ in saved:
        setattr(o, n, make(n))
    try:
        yield
    finally:
        for o, n, f in saved:
            setattr(o, n, f)
