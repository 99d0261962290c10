"""Reference import path `lavila.models.timesformer` -> MI355X-native implementation (lavila_amd.timesformer)."""
import sys as _sys

import lavila_amd.timesformer as _impl

_sys.modules[__name__] = _impl
