"""ctypes binding of liblavila_hip.so (the C ABI declared in include/lavila_hip.h).

There is NO fallback: if the library is missing or a tensor is not on an MI355X device the call raises.
PyTorch is used for device memory and streams only (tensor.data_ptr(), torch.cuda.current_stream()).
"""
import ctypes
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'lib', 'liblavila_hip.so')

LVL_F32, LVL_BF16 = 0, 1
ATTN_SPACE, ATTN_TIME = 0, 1
EPI_BIAS, EPI_BIAS_QUICKGELU, EPI_QUICKGELU_BWD, EPI_BIAS_RESIDUAL, EPI_BIAS_QUICKGELU_DERIV, EPI_MUL_AUX_COLSUM = 0, 1, 2, 3, 4, 5
ACT_GELU_NEW, ACT_SQRELU = 0, 1

_c = ctypes
_P, _I, _L, _F = _c.c_void_p, _c.c_int, _c.c_int64, _c.c_float

# name -> (restype, argtypes); must list every function of include/lavila_hip.h
SIGNATURES = {
    'lvl_version': (_c.c_char_p, []),
    'lvl_set_compute_units': (_I, [_I]),
    'lvl_debug_late_workgroups': (_I, [_I]),
    'lvl_last_error': (_c.c_char_p, []),
    'lvl_workspace_floats': (_L, [_c.c_char_p, _L, _L]),
    'lvl_layernorm_fwd': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _F, _I, _P]),
    'lvl_layernorm_bwd': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _P]),
    'lvl_bias_quickgelu_fwd': (_I, [_P, _P, _P, _L, _I, _I, _P]),
    'lvl_bias_quickgelu_bwd': (_I, [_P, _P, _P, _P, _P, _P, _L, _I, _I, _P]),
    'lvl_patchify': (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    'lvl_embed_tokens_fwd': (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    'lvl_divided_attn_fwd': (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    'lvl_divided_attn_bwd': (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    'lvl_attention_fast_path': (_I, [_I, _I, _I, _I]),
    'lvl_attention_fast_path_f32': (_I, [_I, _I, _I, _I]),
    'lvl_debug_f32_generic': (_I, [_I]),
    'lvl_debug_space_stream': (_I, [_I]),
    'lvl_debug_stream_variant': (_I, [_I]),
    'lvl_set_fp8_qk': (_I, [_I]),
    'lvl_debug_generic_attention_calls': (_I, [_I]),
    'lvl_causal_attn_fwd': (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    'lvl_causal_attn_bwd': (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    'lvl_clip_loss_fwd': (_I, [_P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _I, _P]),
    'lvl_clip_loss_bwd': (_I, [_P, _P, _P, _P, _P, _F, _I, _I, _I, _I, _I, _P, _P, _I, _P]),
    'lvl_ssl_clip_loss_fwd': (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _I, _P]),
    'lvl_ssl_clip_loss_bwd': (_I, [_P, _P, _P, _P, _P, _P, _F, _I, _I, _I, _I, _P, _P, _I, _P]),
    'lvl_linear_tn': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _I, _P]),
    'lvl_linear_wgrad': (_I, [_P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _P]),
    'lvl_cast_transpose': (_I, [_P, _P, _P, _I, _I, _P]),
    'lvl_cast_transpose_multi': (_I, [_P, _I, _L, _P]),
    'lvl_split_bf16x3': (_I, [_P, _P, _L, _I, _L, _L, _L, _I, _P]),
    'lvl_cls_attn_fwd': (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    'lvl_cls_attn_bwd': (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    'lvl_mq_cross_attn_fwd': (_I, [_P, _L, _P, _P, _I, _I, _I, _I, _I, _P]),
    'lvl_qkv_bias_grad': (_I, [_P, _P, _P, _P, _L, _I, _I, _P]),
    'lvl_divided_attn_bwd_bias_ws': (_L, [_I, _I, _I, _I, _I, _I]),
    'lvl_divided_attn_bwd_bias': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    'lvl_vec_mat_f32': (_I, [_P, _P, _P, _I, _I, _P]),
    'lvl_embed_tokens_bwd_ws': (_L, [_I, _I, _I]),
    'lvl_embed_tokens_bwd': (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    'lvl_text_embed_fwd': (_I, [_P, _L, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    'lvl_text_embed_bwd_ws': (_L, [_I, _I, _I]),
    'lvl_text_embed_bwd': (_I, [_P, _P, _L, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    'lvl_debug_time_bwd_rider': (_I, [_I]),
    'lvl_linear_skinny': (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    'lvl_linear_skinny_f32c': (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    'lvl_linear_skinny_ln': (_I, [_P, _P, _P, _P, _P, _F, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    'lvl_debug_skinny_variant': (_I, [_I]),
    'lvl_debug_cross_attn_waves': (_I, [_I]),
    'lvl_sample_max_vocab': (_I, []),
    'lvl_sample_next_token': (_I, [_P, _L, _I, _I, _F, _I, _F, _P, _P, _L, _P, _P, _P, _P, _P]),
    'lvl_gpt2_embed': (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    'lvl_gated_add_layernorm': (_I, [_P, _P, _P, _P, _P, _F, _P, _P, _I, _I, _I, _P]),
    'lvl_act_inplace': (_I, [_P, _L, _I, _I, _P]),
    'lvl_decode_self_attn': (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    'lvl_cross_attn_rows_fwd': (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
}

_lib = None
_lock = threading.Lock()


class HipExtensionError(RuntimeError):
    pass


def lib():
    """Loads liblavila_hip.so once. Raises HipExtensionError (never falls back) if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise HipExtensionError(
                    f'{LIB_PATH} is missing: build it with `python -m lavila_amd.build` '
                    '(or __graft_entry__.build()); lavila_amd has no CPU / eager fallback')
            try:
                handle = ctypes.CDLL(LIB_PATH)
            except OSError as e:
                raise HipExtensionError(f'cannot load {LIB_PATH}: {e}') from e
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(handle, name)
                fn.restype = res
                fn.argtypes = args
            cus = os.environ.get('LAVILA_COMPUTE_UNITS')        # CUs for the persistent GEMM kernels (see the header)
            if cus and handle.lvl_set_compute_units(int(cus)) != 0:
                raise HipExtensionError('LAVILA_COMPUTE_UNITS: ' + handle.lvl_last_error().decode(errors='replace'))
            _lib = handle
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().lvl_last_error().decode(errors='replace')
        raise HipExtensionError(f'{what} failed (status {rc}): {msg}')


def dtype_code(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return LVL_F32
    if t.dtype == torch.bfloat16:
        return LVL_BF16
    raise HipExtensionError(
        f'lavila_amd kernels take float32 or bfloat16 activations, got {t.dtype} '
        '(fp16 autocast is remapped to bf16 by the model wrappers; see INTEGRATION.md)')


def require_device(*tensors):
    """Every kernel is queued on the CURRENT device's current stream (stream_ptr): tensors must live there."""
    cur = None
    for t in tensors:
        if t is None:
            continue
        if t.is_cuda:
            if cur is None:
                cur = torch.cuda.current_device()
            if t.device.index != cur:
                raise HipExtensionError(
                    f'lavila_amd: tensor on {t.device} but the current device is cuda:{cur}; call '
                    'torch.cuda.set_device() (one process per GPU) before running the model')
        if not t.is_cuda:
            raise HipExtensionError(
                'lavila_amd: tensor on %s -- the HIP kernels need a ROCm device tensor; there is no CPU '
                'fallback (the CPU oracle under oracle/ is test infrastructure only)' % t.device)
        if not t.is_contiguous():
            raise HipExtensionError('lavila_amd: non-contiguous tensor passed to a HIP kernel')


def ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def workspace(op: str, rows: int, cols: int, device) -> torch.Tensor:
    n = lib().lvl_workspace_floats(op.encode(), rows, cols)
    if n < 0:
        raise HipExtensionError(f'unknown workspace op {op}')
    return torch.empty(max(int(n), 1), dtype=torch.float32, device=device)
