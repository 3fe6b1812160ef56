"""MI355X-native implementation of the Transform-and-Tell caption hot path.

Directory name `transform-and-tell_amd`; import it as `tell_amd` (alias module at
the repository root).  The arithmetic lives in csrc/*.hip (gfx950) behind the C
ABI of include/tell_hip.h; this package is the Python host side that mirrors the
reference's plugin surface (`tell.models`, `tell.modules`, registrable names,
constructor arguments, state_dict keys).
"""
from . import hip, runtime, streams  # noqa: F401
from .runtime import (compute_dtype, manual_seed, set_compute_dtype,  # noqa: F401
                      bump_weights_epoch)

__all__ = ['hip', 'runtime', 'compute_dtype', 'set_compute_dtype', 'manual_seed', 'bump_weights_epoch']
