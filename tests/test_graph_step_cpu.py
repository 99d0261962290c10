"""CPU: host logic of lavila_amd.graph_step (the capture itself needs a device: tests/test_gpu_graph_step.py)."""
import pytest
import torch

from lavila_amd import models
from lavila_amd.graph_step import GraphedTrainStep


def test_fixed_text_length_nests_and_restores():
    assert getattr(models._fixed_len, 'value', None) is None
    with models.fixed_text_length(24):
        assert models._fixed_len.value == 24
        with models.fixed_text_length(None):
            assert models._fixed_len.value is None
        assert models._fixed_len.value == 24
    assert models._fixed_len.value is None


def test_caption_bucket_is_taken_from_host_tokens_and_device_tokens_get_the_full_context():
    step = GraphedTrainStep.__new__(GraphedTrainStep)          # host logic only: no device buffers
    step.context, step.text_bucket = 77, 8
    tokens = torch.zeros(3, 77, dtype=torch.long)
    tokens[0, 4], tokens[1, 20], tokens[2, 9] = 999, 999, 999    # EOT (highest id) at positions 4, 20, 9
    assert step.caption_length(tokens) == 24                     # 21 positions -> next multiple of 8
    tokens[1, 20], tokens[1, 76] = 0, 999
    assert step.caption_length(tokens) == 77
    step.text_bucket = 1
    tokens[1, 76], tokens[1, 30] = 0, 999
    assert step.caption_length(tokens) == 31


@pytest.mark.skipif(torch.cuda.is_available(), reason='CPU-only behaviour')
def test_graphed_step_is_loud_without_a_device():
    lin = torch.nn.Linear(4, 4)
    with pytest.raises(RuntimeError, match='HIP device'):
        GraphedTrainStep(lin, None, torch.optim.AdamW(lin.parameters()), (1, 3, 1, 8, 8), (1, 77), 'cpu')


def test_distributed_data_parallel_models_are_refused():
    """Measured on the GPU box (one-rank RCCL group): the process-group watchdog aborts the process during a capture."""
    ddp = torch.nn.parallel.DistributedDataParallel.__new__(torch.nn.parallel.DistributedDataParallel)
    with pytest.raises(NotImplementedError, match='DistributedDataParallel'):
        GraphedTrainStep(ddp, None, None, (1, 3, 1, 8, 8), (1, 77), 'cpu')


def test_caption_bound_from_host_tokens():
    """models.caption_bound: 1 + the largest EOT position of a HOST token tensor, rounded up to the bucket, capped at the
    context; device tensors are refused (reading them is the sync the helper exists to avoid)."""
    import pytest
    import torch
    from lavila.models import models
    tokens = torch.zeros(3, 77, dtype=torch.long)
    tokens[0, 20], tokens[1, 4], tokens[2, 11] = 49407, 49407, 49407
    assert models.caption_bound(tokens) == 24
    assert models.caption_bound(tokens, bucket=1) == 21
    tokens[1, 4], tokens[1, 76] = 7, 49407           # a caption that fills the context
    assert models.caption_bound(tokens) == 77
    with models.fixed_text_length(models.caption_bound(tokens[:1])):
        assert models._fixed_len.value == 24
    if torch.cuda.is_available():
        with pytest.raises(ValueError):
            models.caption_bound(tokens.cuda())
