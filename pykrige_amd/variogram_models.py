"""The reference's variogram model functions by name (variogram_models.py:25-81): ``f(m, d)`` with
``m = [slope, nugget]`` / ``[scale, exponent, nugget]`` / ``[psill, range, nugget]``.  On the execute() path the
device evaluates its own functors (mik_kernels.h `vario<MODEL>`); these host twins exist so that code written
against ``pykrige.variogram_models`` (and ``core._krige`` / ``_find_statistics``, which take such a function and
dispatch on its ``__name__`` exactly like lib/variogram_models.pyx:6-22) keeps working."""
from . import core as _core


def linear_variogram_model(m, d):
    return _core.variogram_value("linear", m, d)


def power_variogram_model(m, d):
    return _core.variogram_value("power", m, d)


def gaussian_variogram_model(m, d):
    return _core.variogram_value("gaussian", m, d)


def exponential_variogram_model(m, d):
    return _core.variogram_value("exponential", m, d)


def spherical_variogram_model(m, d):
    return _core.variogram_value("spherical", m, d)


def hole_effect_variogram_model(m, d):
    return _core.variogram_value("hole-effect", m, d)


MODEL_OF_FUNCTION = {
    "linear_variogram_model": "linear", "power_variogram_model": "power", "gaussian_variogram_model": "gaussian",
    "exponential_variogram_model": "exponential", "spherical_variogram_model": "spherical",
    "hole_effect_variogram_model": "hole-effect",
}
