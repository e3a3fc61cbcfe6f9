"""pykrige_amd -- the PyKrige execute() hot path on AMD MI355X (gfx950).

Drop-in for ``OrdinaryKriging / UniversalKriging / OrdinaryKriging3D / UniversalKriging3D .execute``:
kriging-matrix assembly, dense inverse, per-point right-hand sides and the z / sigma^2 contraction run
as hand-written HIP kernels in ``libmikrige.so`` (C ABI: include/mikrige.h), called through ctypes.
"""
from .kriging import OrdinaryKriging, OrdinaryKriging3D, UniversalKriging, UniversalKriging3D  # noqa: F401
from . import _lib, core, variogram_models  # noqa: F401
from ._lib import set_devices  # noqa: F401  (single-process multi-GPU: handles span n GPUs; MIK_NGPU does the same)
from . import kriging_tools as kt  # noqa: F401  (the reference's alias)
from . import ok, ok3d, uk, uk3d  # noqa: F401,E402  (`import pykrige; pykrige.ok.OrdinaryKriging` works upstream: pykrige/__init__.py:44-48 binds the submodules)

__all__ = ["OrdinaryKriging", "UniversalKriging", "OrdinaryKriging3D", "UniversalKriging3D", "kt", "ok", "uk", "ok3d", "uk3d", "kriging_tools", "__version__"]
__version__ = "0.1.0"
