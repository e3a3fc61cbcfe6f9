"""TEST / BASELINE INFRASTRUCTURE.  Loads the REAL reference's native loop (lib/cok.pyx `_c_exec_loop`,
compiled by oracle/build_ref.sh into oracle/_ref/pykrige_lib/) without the reference's Python package:
the two extension modules only need to find each other as `pykrige.lib.*`, so a bare namespace is
registered for them.  Used by bench.py's cpu_baseline leg (kind "reference") and by tests."""
import glob
import os
import sys
import types

import numpy as np
from scipy.spatial.distance import cdist

from . import kriging_oracle as ko

_REF_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "pykrige_lib")
_VARIOGRAM_NAMES = {  # lib/variogram_models.pyx:6-22 dispatches on the Python function's __name__
    "linear": "linear_variogram_model", "power": "power_variogram_model", "gaussian": "gaussian_variogram_model",
    "exponential": "exponential_variogram_model", "spherical": "spherical_variogram_model",
}


def available():
    return bool(glob.glob(os.path.join(_REF_DIR, "cok*.so")))


def load_cok():
    if "pykrige.lib.cok" in sys.modules:
        return sys.modules["pykrige.lib.cok"]
    if not available():
        raise ImportError("oracle/_ref/pykrige_lib/cok*.so not built (run oracle/build_ref.sh where /root/reference exists)")
    if "pykrige" not in sys.modules:
        pk = types.ModuleType("pykrige")
        pk.__path__ = []
        sys.modules["pykrige"] = pk
    if "pykrige.lib" not in sys.modules:
        lib = types.ModuleType("pykrige.lib")
        lib.__path__ = [_REF_DIR]
        sys.modules["pykrige.lib"] = lib
    elif _REF_DIR not in list(sys.modules["pykrige.lib"].__path__):
        sys.modules["pykrige.lib"].__path__.append(_REF_DIR)
    import pykrige.lib.cok as cok  # noqa

    return cok


def c_backend(st, pts_adj):
    """backend='C' of OrdinaryKriging.execute (ok.py:903-927, 989, 1002-1005) on adjusted points:
    cdist -> _c_exec_loop(a, bd, mask, n, pars).  Returns (z, sigma^2, seconds spent in the native loop)."""
    import time

    cok = load_cok()
    name = _VARIOGRAM_NAMES[st.model]
    fn = lambda m, d: ko.variogram(st.model, m, d)  # noqa: E731  (only its __name__ is read by the C side)
    fn.__name__ = name
    a = ko.kriging_matrix(st)
    bd = cdist(pts_adj, st.coords_adj, "euclidean")
    pars = dict(Z=st.values, eps=ko.EPS, variogram_model_parameters=np.asarray(st.params, dtype=np.float64),
                variogram_function=fn, exact_values=st.exact_values, pseudo_inv=False, pseudo_inv_type="pinv")
    t0 = time.perf_counter()
    z, ss = cok._c_exec_loop(a, bd, np.zeros(pts_adj.shape[0], dtype="int8"), st.n, pars)
    return np.asarray(z), np.asarray(ss), time.perf_counter() - t0


def c_backend_moving_window(st, pts_adj, n_closest_points):
    """backend='C' with n_closest_points (ok.py:929-986): cKDTree.query -> _c_exec_loop_moving_window(a, bd, mask, bd_idx,
    n, pars).  Returns (z, sigma^2, seconds in cKDTree + the native loop, seconds in the native loop alone)."""
    import time

    from scipy.spatial import cKDTree

    cok = load_cok()
    fn = lambda m, d: ko.variogram(st.model, m, d)  # noqa: E731
    fn.__name__ = _VARIOGRAM_NAMES[st.model]
    a = ko.kriging_matrix(st)
    pars = dict(Z=st.values, eps=ko.EPS, variogram_model_parameters=np.asarray(st.params, dtype=np.float64),
                variogram_function=fn, exact_values=st.exact_values, pseudo_inv=False, pseudo_inv_type="pinv")
    t0 = time.perf_counter()
    tree = cKDTree(st.coords_adj)
    bd, bd_idx = tree.query(pts_adj, k=int(n_closest_points), eps=0.0)
    t1 = time.perf_counter()
    z, ss = cok._c_exec_loop_moving_window(a, np.ascontiguousarray(bd), np.zeros(pts_adj.shape[0], dtype="int8"),
                                           np.ascontiguousarray(bd_idx.astype("long")), st.n, pars)
    t2 = time.perf_counter()
    return np.asarray(z), np.asarray(ss), t2 - t0, t2 - t1
