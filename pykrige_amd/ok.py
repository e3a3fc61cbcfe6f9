"""Module-path alias of the reference's `pykrige.ok` (`ok.py`): `from pykrige_amd.ok import OrdinaryKriging`."""
from .kriging import OrdinaryKriging  # noqa: F401

__all__ = ["OrdinaryKriging"]
