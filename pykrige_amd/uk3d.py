"""Module-path alias of the reference's `pykrige.uk3d` (`uk3d.py`): `from pykrige_amd.uk3d import UniversalKriging3D`."""
from .kriging import UniversalKriging3D  # noqa: F401

__all__ = ["UniversalKriging3D"]
