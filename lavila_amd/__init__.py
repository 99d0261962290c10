"""lavila_amd -- MI355X-native (gfx950) implementation of the LaViLa dual-encoder pretraining hot path.

Layout: csrc/ (hand-written HIP kernels + the C ABI of include/lavila_hip.h), _cabi.py (ctypes binding),
ops.py (autograd pairing of forward/backward kernels) and the host-side mirror of the reference interface:
models.py, timesformer.py, openai_model.py, loss.py, distributed_utils.py, utils.py
(= lavila/models/*.py of facebookresearch/LaViLa). The top-level `lavila` package re-exports them under the
reference's import paths so main_pretrain.py / eval_zeroshot.py run unchanged.
"""
__version__ = '0.1.0'
