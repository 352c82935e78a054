"""rocket_path_amd -- MI355X-native batched interior-point trajectory step.

Only what the hot path needs: csrc/ (HIP kernels, the C ABI, the C++ Problem plug-in) and
this thin Python mirror of the same interface.  (The package directory uses '_' where the
project name has '-': Python identifiers cannot contain a hyphen.)
"""
from . import capi, problems  # noqa: F401
from .batch import Batch, Pipeline  # noqa: F401
from .capi import (DTYPE_F32, DTYPE_F32_STATE, DTYPE_F64, ST_CONVERGED, ST_INFEASIBLE, ST_MAXITER, ST_NONFINITE, ST_STALLED, ST_WRONG_WAY,  # noqa: F401
                   VARIANT_F3, VARIANT_F4, RpError, device_count, device_id, load_library)
