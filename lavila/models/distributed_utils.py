"""Reference import path `lavila.models.distributed_utils` -> MI355X-native implementation (lavila_amd.distributed_utils)."""
import sys as _sys

import lavila_amd.distributed_utils as _impl

_sys.modules[__name__] = _impl
