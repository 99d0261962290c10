"""Reference import path `lavila.models.models` -> MI355X-native implementation (lavila_amd.models)."""
import sys as _sys

import lavila_amd.models as _impl

_sys.modules[__name__] = _impl
