"""Reference import path `lavila.models.openai_model` -> MI355X-native implementation (lavila_amd.openai_model)."""
import sys as _sys

import lavila_amd.openai_model as _impl

_sys.modules[__name__] = _impl
