"""Module-path alias of the reference's `pykrige.ok3d` (`ok3d.py`): `from pykrige_amd.ok3d import OrdinaryKriging3D`."""
from .kriging import OrdinaryKriging3D  # noqa: F401

__all__ = ["OrdinaryKriging3D"]
