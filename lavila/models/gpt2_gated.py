"""Reference import path `lavila.models.gpt2_gated` -> MI355X-native implementation (lavila_amd.gpt2_gated), inference only."""
import sys as _sys

import lavila_amd.gpt2_gated as _impl

_sys.modules[__name__] = _impl
