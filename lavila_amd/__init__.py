"""lavila_amd -- MI355X-native (gfx950) implementation of the LaViLa dual-encoder pretraining hot path.

Layout: csrc/ (hand-written HIP kernels + the C ABI of include/lavila_hip.h), _cabi.py (ctypes binding),
ops.py (autograd pairing of forward/backward kernels) and the host-side mirror of the reference interface:
models.py, timesformer.py, openai_model.py, loss.py, distributed_utils.py, utils.py
(= lavila/models/*.py of facebookresearch/LaViLa). The top-level `lavila` package re-exports them under the
reference's import paths so main_pretrain.py / eval_zeroshot.py run unchanged.
"""
__version__ = '0.1.0'


def _reserve_hardware_queues():
    """One rank of a multi-GPU job: ask the HIP runtime for 8 hardware queues instead of its default 4.

    HIP multiplexes all streams of a process onto GPU_MAX_HW_QUEUES hardware queues (default 4). The moment an RCCL
    communicator exists its streams take the queue of the text tower's side stream, the two towers serialise and the step
    is 3.7-4.5 % slower with nothing on the wire (profiles/r05_one_rank_group_bisect.txt; 8 queues: 0.0 %). The runtime
    reads the variable when it initialises, so it has to be in the environment before the first device call -- the import
    of the drop-in (main_pretrain.py:28 imports lavila.models before it touches the GPU, :151-183) is the last point we
    own. Only when the launcher announced more than one rank (torchrun exports WORLD_SIZE), never over a value the user
    set; if the runtime is already up the setting cannot take effect any more and we say so once."""
    import os
    import sys
    try:
        world = int(os.environ.get('WORLD_SIZE', '1'))
    except ValueError:
        world = 1
    if world <= 1 or 'GPU_MAX_HW_QUEUES' in os.environ or os.environ.get('LAVILA_HW_QUEUES', '1') == '0':
        return None
    torch = sys.modules.get('torch')
    if torch is not None and torch.cuda.is_initialized():
        import warnings
        warnings.warn('lavila_amd: WORLD_SIZE > 1 but the HIP runtime was initialised before lavila_amd was imported, so '
                      'GPU_MAX_HW_QUEUES=8 cannot be applied any more: RCCL\'s streams will share the text tower\'s '
                      'hardware queue (step +4 %). Export GPU_MAX_HW_QUEUES=8 in the launcher (INTEGRATION.md section 4)')
        return False
    os.environ['GPU_MAX_HW_QUEUES'] = '8'
    return True


HW_QUEUES_RESERVED = _reserve_hardware_queues()
