"""scikit-learn side callers of the execute path (reference: compat.py:97-307).

`Krige` is the estimator the reference's GridSearchCV examples, `RegressionKriging` (rk.py:151-166) and
`ClassificationKriging` (ck.py:172-192) drive: fit() builds one of the four kriging classes, predict()/execute() call
its execute(style="points", backend="loop", n_closest_points=...) -- i.e. on this package the device moving-window
path (mik_predict_moving_window) for the ordinary methods and the dense path (mik_predict) for the universal ones.
"""
import numpy as np

from .kriging import OrdinaryKriging, OrdinaryKriging3D, UniversalKriging, UniversalKriging3D

try:
    from sklearn.base import BaseEstimator, ClassifierMixin, RegressorMixin
    from sklearn.model_selection import train_test_split  # noqa: F401  (kept importable from here, as upstream keeps it: compat.py:11-13)

    SKLEARN_INSTALLED = True
except ImportError:  # same degradation as the reference (compat.py:13-25): the classes exist, the checks fail
    SKLEARN_INSTALLED = False
    train_test_split = None

    class RegressorMixin:
        pass

    class ClassifierMixin:
        pass

    class BaseEstimator:
        pass


krige_methods = {"ordinary": OrdinaryKriging, "universal": UniversalKriging,
                 "ordinary3d": OrdinaryKriging3D, "universal3d": UniversalKriging3D}
threed_krige = ("ordinary3d", "universal3d")

# which constructor keywords each method receives beyond the common ones (compat.py:37-74)
_ANISO_2D = ("anisotropy_scaling", "anisotropy_angle")
_ANISO_3D = ("anisotropy_scaling_y", "anisotropy_scaling_z", "anisotropy_angle_x", "anisotropy_angle_y", "anisotropy_angle_z")
krige_methods_kws = {
    "ordinary": list(_ANISO_2D + ("enable_statistics", "coordinates_type")),
    "universal": list(_ANISO_2D + ("drift_terms", "point_drift", "external_drift", "external_drift_x", "external_drift_y",
                                   "functional_drift")),
    "ordinary3d": list(_ANISO_3D),
    "universal3d": list(_ANISO_3D + ("drift_terms", "functional_drift")),
}


class SklearnException(Exception):
    """scikit-learn is needed and missing."""


def validate_method(method):
    if method not in krige_methods:
        raise ValueError("Kriging method must be one of {}".format(krige_methods.keys()))


def validate_sklearn():
    if not SKLEARN_INSTALLED:
        raise SklearnException("sklearn needs to be installed in order to use this module")


class Krige(RegressorMixin, BaseEstimator):
    """scikit-learn estimator around the four kriging classes; parameters as compat.py:143-185."""

    def __init__(self, method="ordinary", variogram_model="linear", nlags=6, weight=False, n_closest_points=10,
                 verbose=False, exact_values=True, pseudo_inv=False, pseudo_inv_type="pinv", variogram_parameters=None,
                 variogram_function=None, anisotropy_scaling=(1.0, 1.0), anisotropy_angle=(0.0, 0.0, 0.0),
                 enable_statistics=False, coordinates_type="euclidean", drift_terms=None, point_drift=None,
                 ext_drift_grid=(None, None, None), functional_drift=None):
        validate_method(method)
        given = dict(locals())
        for name in ("self", "__class__"):
            given.pop(name, None)
        for name, value in given.items():  # scikit-learn's get_params / clone contract: stored under the argument names
            setattr(self, name, value)
        self.model = None  # set by fit()

    def _method_specific(self):
        sc, an, ext = self.anisotropy_scaling, self.anisotropy_angle, self.ext_drift_grid
        pool = {
            "anisotropy_scaling": sc[0], "anisotropy_angle": an[0],
            # the reference hands scaling[0] to y and scaling[1] to z (compat.py:210-211)
            "anisotropy_scaling_y": sc[0], "anisotropy_scaling_z": sc[1],
            "anisotropy_angle_x": an[0], "anisotropy_angle_y": an[1], "anisotropy_angle_z": an[2],
            "enable_statistics": self.enable_statistics, "coordinates_type": self.coordinates_type,
            "drift_terms": self.drift_terms, "point_drift": self.point_drift,
            "external_drift": ext[0], "external_drift_x": ext[1], "external_drift_y": ext[2],
            "functional_drift": self.functional_drift,
        }
        return {k: pool[k] for k in krige_methods_kws[self.method]}

    def _dimensionality_check(self, x, ext=""):
        want = 3 if self.method in threed_krige else 2
        if x.shape[1] != want:
            raise ValueError("%dd krige can use only %dd points" % (want, want))
        return {name + ext: x[:, i] for i, name in enumerate("xyz"[:want])}

    def fit(self, x, y, *args, **kwargs):
        """x: (N, 2) or (N, 3) station coordinates, y: (N,) values."""
        kw = self._dimensionality_check(x)
        kw["val" if self.method in threed_krige else "z"] = y
        for name in ("variogram_model", "variogram_parameters", "variogram_function", "nlags", "weight", "verbose",
                     "exact_values", "pseudo_inv", "pseudo_inv_type"):
            kw[name] = getattr(self, name)
        kw.update(self._method_specific())
        self.model = krige_methods[self.method](**kw)

    def predict(self, x, *args, **kwargs):
        """Kriged values at x ((N, 2) / (N, 3) points)."""
        if not self.model:
            raise Exception("Not trained. Train first")
        return self.execute(self._dimensionality_check(x, ext="points"), *args, **kwargs)[0]

    def execute(self, points, *args, **kwargs):
        """points: dict xpoints/ypoints[/zpoints]; returns (prediction, variance) (compat.py:269-291)."""
        call = dict(points, style="points", backend="loop")
        call.update(kwargs)
        if type(self.model) in (OrdinaryKriging, OrdinaryKriging3D):  # exact types: here the universal classes derive from these
            call["n_closest_points"] = self.n_closest_points
        else:
            print("n_closest_points will be ignored for UniversalKriging")
        return self.model.execute(**call)


def check_sklearn_model(model, task="regression"):
    """The wrapped learner has to be a scikit-learn estimator of the right kind (compat.py:294-307)."""
    mixin = {"regression": RegressorMixin, "classification": ClassifierMixin}.get(task)
    if mixin is not None and not (isinstance(model, BaseEstimator) and isinstance(model, mixin)):
        raise RuntimeError("Needs to supply an instance of a scikit-learn %s class." % task)
