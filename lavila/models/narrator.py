"""Reference import path `lavila.models.narrator` -> MI355X-native implementation (lavila_amd.narrator), inference only."""
import sys as _sys

import lavila_amd.narrator as _impl

_sys.modules[__name__] = _impl
