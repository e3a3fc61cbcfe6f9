"""Module-path alias of the reference's `pykrige.uk` (`uk.py`): `from pykrige_amd.uk import UniversalKriging`."""
from .kriging import UniversalKriging  # noqa: F401

__all__ = ["UniversalKriging"]
