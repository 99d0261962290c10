"""Reference import path `lavila.models.utils` -> MI355X-native implementation (lavila_amd.utils)."""
import sys as _sys

import lavila_amd.utils as _impl

_sys.modules[__name__] = _impl
