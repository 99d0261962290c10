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


_REF = None


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
