"""pykrige_amd -- the PyKrige execute() hot path on AMD MI355X (gfx950).

Drop-in for ``OrdinaryKriging / UniversalKriging / OrdinaryKriging3D / UniversalKriging3D .execute``:
kriging-matrix assembly, dense inverse, per-point right-hand sides and the z / sigma^2 contraction run
as hand-written HIP kernels in ``libmikrige.so`` (C ABI: include/mikrige.h), called through ctypes.
"""
from .kriging import OrdinaryKriging, OrdinaryKriging3D, UniversalKriging, UniversalKriging3D  # noqa: F401
from . import _lib, core, variogram_models  # noqa: F401
from ._lib import set_devices  # noqa: F401  (single-process multi-GPU: handles span n GPUs; MIK_NGPU does the same)
from . import kriging_tools as kt  # noqa: F401  (the reference's alias)

__all__ = ["OrdinaryKriging", "UniversalKriging", "OrdinaryKriging3D", "UniversalKriging3D"]
__version__ = "0.1.0"
