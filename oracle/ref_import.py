"""TEST INFRASTRUCTURE ONLY -- imports the *real* reference (facebookresearch/LaViLa at
/root/reference) on CPU so that (a) golden vectors can be generated from it and (b) the
restatement in oracle/oracle.py can be pinned against it.

/root/reference only exists in the build container, never on the GPU box; nothing under
tests/ -m gpu, smoke() or bench.py may call into this module.

The reference imports three things that are absent offline:
  * timm.models.layers.{DropPath,to_2tuple,trunc_normal_}   (lavila/models/timesformer.py:31)
  * lavila.models.{gpt2_gated,narrator,openai_clip}          (lavila/models/models.py:15-18)
They are replaced by minimal in-memory stand-ins *before* importing lavila.models.models.
"""
import importlib
import importlib.machinery
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("LAVILA_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "lavila", "models"))


def _mod(name):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, loader=None)
    return m


def _install_stubs():
    import torch
    import torch.nn as nn

    # transformers probes timm.__spec__; import it first so that the probe sees "no timm".
    import transformers  # noqa: F401
    from transformers import DistilBertModel, GPT2LMHeadModel  # noqa: F401

    if "timm" not in sys.modules:
        timm = _mod("timm")
        timm_models = _mod("timm.models")
        timm_layers = _mod("timm.models.layers")
        timm_vit = _mod("timm.models.vision_transformer")

        class DropPath(nn.Module):
            def __init__(self, drop_prob=0.0):
                super().__init__()
                self.drop_prob = drop_prob

            def forward(self, x):
                if self.drop_prob == 0.0 or not self.training:
                    return x
                keep = 1.0 - self.drop_prob
                shape = (x.shape[0],) + (1,) * (x.ndim - 1)
                mask = x.new_empty(shape).bernoulli_(keep)
                return x * mask / keep

        def to_2tuple(x):
            return tuple(x) if isinstance(x, (tuple, list)) else (x, x)

        timm_layers.DropPath = DropPath
        timm_layers.to_2tuple = to_2tuple
        timm_layers.trunc_normal_ = torch.nn.init.trunc_normal_
        timm.models = timm_models
        timm_models.layers = timm_layers
        timm_models.vision_transformer = timm_vit
        sys.modules.update({
            "timm": timm, "timm.models": timm_models,
            "timm.models.layers": timm_layers, "timm.models.vision_transformer": timm_vit,
        })

    for name, attrs in (
        ("lavila.models.gpt2_gated", ("GPT2LMHeadModel", "augment_gpt2_config")),
        ("lavila.models.narrator", ("VCLM_HF",)),
        ("lavila.models.openai_clip", ("load",)),
    ):
        if name not in sys.modules:
            m = _mod(name)
            for a in attrs:
                setattr(m, a, None)
            sys.modules[name] = m


def _install_transformers_4_27_names():
    """lavila/models/gpt2_gated.py:36-48 and narrator.py:16 import names of transformers 4.27 that 5.x dropped
    (SequenceSummary, head pruning helpers, model_parallel_utils, the docstring decorators, BeamSearchScorer). Except for
    BeamSearchScorer (restated in oracle/beam_scorer.py for the beam-search goldens) none of
    them is on the path the goldens exercise (GPT2LMHeadModel.forward, VCLM_HF.forward / generate), so inert stand-ins
    are installed under the old names before the unmodified reference source is imported. `get_head_mask` /
    `invert_attention_mask` (PreTrainedModel methods the forward calls, gpt2_gated.py:884,889) are restated from their
    4.27 definitions when the installed base class no longer has them."""
    import torch
    import torch.nn as nn
    import transformers
    import transformers.modeling_utils as mu
    import transformers.pytorch_utils as pu
    import transformers.utils as tu

    if not hasattr(mu, "SequenceSummary"):
        class SequenceSummary(nn.Module):            # only GPT2DoubleHeadsModel builds one
            def __init__(self, config):
                super().__init__()
        mu.SequenceSummary = SequenceSummary

    def _unused(*a, **k):
        raise RuntimeError("stand-in for a transformers 4.27 helper that the golden path never calls")

    for name in ("find_pruneable_heads_and_indices", "prune_conv1d_layer"):
        if not hasattr(pu, name):
            setattr(pu, name, _unused)

    def _decorator(*a, **k):
        return lambda fn: fn

    for name in ("add_code_sample_docstrings", "add_start_docstrings", "add_start_docstrings_to_model_forward",
                 "replace_return_docstrings"):
        if not hasattr(tu, name):
            setattr(tu, name, _decorator)
    if "transformers.utils.model_parallel_utils" not in sys.modules:
        try:
            importlib.import_module("transformers.utils.model_parallel_utils")
        except Exception:
            m = _mod("transformers.utils.model_parallel_utils")
            m.assert_device_map = _unused
            m.get_device_map = _unused
            sys.modules["transformers.utils.model_parallel_utils"] = m
            tu.model_parallel_utils = m
    if not hasattr(transformers, "BeamSearchScorer"):
        # beam_sample / group_beam_search (narrator.py:149-366): the 4.27 class is restated in oracle/beam_scorer.py
        # (parity unpinned against the absent dependency; the reference's own loops around it run unmodified)
        from oracle.beam_scorer import BeamSearchScorer
        transformers.BeamSearchScorer = BeamSearchScorer

    base = mu.PreTrainedModel
    if not hasattr(base, "get_head_mask"):
        def get_head_mask(self, head_mask, num_hidden_layers, is_attention_chunked=False):
            if head_mask is not None:
                raise RuntimeError("head masks are not on the golden path")
            return [None] * num_hidden_layers
        base.get_head_mask = get_head_mask
    if not hasattr(base, "invert_attention_mask"):
        def invert_attention_mask(self, encoder_attention_mask):
            # transformers 4.27 modeling_utils.ModuleUtilsMixin.invert_attention_mask
            if encoder_attention_mask.dim() == 3:
                ext = encoder_attention_mask[:, None, :, :]
            else:
                ext = encoder_attention_mask[:, None, None, :]
            ext = ext.to(dtype=self.dtype)
            return (1.0 - ext) * torch.finfo(self.dtype).min
        base.invert_attention_mask = invert_attention_mask


_REF = None
_REF_NARRATOR = None


def load_reference():
    """Returns a namespace with the reference's hot-path modules (unmodified source)."""
    global _REF
    if _REF is not None:
        return _REF
    if not reference_available():
        raise RuntimeError("reference not mounted at %s" % REFERENCE_ROOT)
    _install_stubs()
    # The repo ships its own `lavila` shim package (drop-in import path). To import the real
    # reference we temporarily put REFERENCE_ROOT first and purge any `lavila*` modules.
    saved = {k: v for k, v in sys.modules.items() if k == "lavila" or k.startswith("lavila.")}
    stubs = {k: v for k, v in saved.items()
             if k in ("lavila.models.gpt2_gated", "lavila.models.narrator", "lavila.models.openai_clip")}
    for k in saved:
        del sys.modules[k]
    sys.modules.update(stubs)
    # `lavila` is a namespace package in both trees (no __init__.py): the first sys.path entry that holds
    # lavila/models/<module>.py wins, so the reference root goes first while its modules are imported
    saved_path = list(sys.path)
    sys.path.insert(0, REFERENCE_ROOT)
    try:
        ns = types.SimpleNamespace()
        ns.timesformer = importlib.import_module("lavila.models.timesformer")
        ns.openai_model = importlib.import_module("lavila.models.openai_model")
        ns.loss = importlib.import_module("lavila.models.loss")
        ns.distributed_utils = importlib.import_module("lavila.models.distributed_utils")
        ns.utils = importlib.import_module("lavila.models.utils")
        ns.coca = importlib.import_module("lavila.models.coca")          # narrator pooling (imports as is)
        ns.models = importlib.import_module("lavila.models.models")
        assert ns.models.__file__.startswith(REFERENCE_ROOT), ns.models.__file__
    finally:
        sys.path[:] = saved_path
        ref_mods = {k: v for k, v in sys.modules.items() if k == "lavila" or k.startswith("lavila.")}
        for k in ref_mods:
            del sys.modules[k]
        sys.modules.update({k: v for k, v in saved.items() if k not in stubs})
    _REF = ns
    return ns


def load_reference_narrator():
    """The reference's narrator side, unmodified source: namespace with `gpt2_gated` (GPT2LMHeadModel,
    augment_gpt2_config; gpt2_gated.py), `narrator` (VCLM_HF; narrator.py) and `coca`, plus everything of
    load_reference(). The module objects are created outside sys.modules' `lavila.*` names (which keep pointing at this
    repository's drop-in package)."""
    global _REF_NARRATOR
    if _REF_NARRATOR is not None:
        return _REF_NARRATOR
    ns = load_reference()
    _install_transformers_4_27_names()
    saved = {k: v for k, v in sys.modules.items() if k == "lavila" or k.startswith("lavila.")}
    for k in saved:
        del sys.modules[k]
    # the narrator imports coca / openai_model / timesformer by their `lavila.models.*` names: hand it the reference
    # modules load_reference() already holds, so that its isinstance checks see the reference towers
    sys.modules.update({
        "lavila.models.coca": ns.coca, "lavila.models.openai_model": ns.openai_model,
        "lavila.models.timesformer": ns.timesformer,
    })
    saved_path = list(sys.path)
    sys.path.insert(0, REFERENCE_ROOT)
    try:
        out = types.SimpleNamespace(**vars(ns))
        out.gpt2_gated = importlib.import_module("lavila.models.gpt2_gated")
        out.narrator = importlib.import_module("lavila.models.narrator")
        assert out.gpt2_gated.__file__.startswith(REFERENCE_ROOT), out.gpt2_gated.__file__
        assert out.narrator.__file__.startswith(REFERENCE_ROOT), out.narrator.__file__
    finally:
        sys.path[:] = saved_path
        for k in [k for k in sys.modules if k == "lavila" or k.startswith("lavila.")]:
            del sys.modules[k]
        sys.modules.update(saved)
    _REF_NARRATOR = out
    return out
