"""Reference import path `lavila.models.loss` -> MI355X-native implementation (lavila_amd.loss)."""
import sys as _sys

import lavila_amd.loss as _impl

_sys.modules[__name__] = _impl
