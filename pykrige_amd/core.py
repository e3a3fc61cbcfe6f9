"""Host-side helpers of the execute() path (NumPy only): the anisotropy transform applied to every
grid point, variogram-parameter normalisation, and the bilinear external-Z lookup.

Restated from the reference's behaviour (paths under /root/reference/src/pykrige):
  core.py:120-193   _adjust_for_anisotropy  (same NumPy calls in the same order, so the adjusted
                    coordinates -- and with them the |d| <= eps coincidence test -- are bit-identical)
  core.py:196-376   _make_variogram_parameter_list
  uk.py:512-628     _calculate_data_point_zscalars
"""
import numpy as np

BOUNDED = ("gaussian", "spherical", "exponential", "hole-effect")
MODELS = ("linear", "power") + BOUNDED


def adjust_for_anisotropy(X, center, scaling, angle):
    """(n, d) coordinates -> anisotropy-adjusted coordinates; d in {2, 3}.  X is not modified."""
    X = np.array(X, dtype=np.float64, copy=True)
    if X.ndim != 2 or X.shape[1] not in (2, 3):
        raise ValueError("coordinates must be (n, 2) or (n, 3)")
    ctr = np.asarray(center, dtype=np.float64)[None, :]
    ang = np.asarray(angle, dtype=np.float64) * np.pi / 180
    X -= ctr
    if X.shape[1] == 2:
        stretch = np.array([[1, 0], [0, scaling[0]]])
        c, s = np.cos(-ang[0]), np.sin(-ang[0])
        rot = np.array([[c, -s], [s, c]])
    else:
        stretch = np.array([[1.0, 0.0, 0.0], [0.0, scaling[0], 0.0], [0.0, 0.0, scaling[1]]])
        cx, sx = np.cos(-ang[0]), np.sin(-ang[0])
        cy, sy = np.cos(-ang[1]), np.sin(-ang[1])
        cz, sz = np.cos(-ang[2]), np.sin(-ang[2])
        rot_x = np.array([[1.0, 0.0, 0.0], [0.0, cx, -sx], [0.0, sx, cx]])
        rot_y = np.array([[cy, 0.0, sy], [0.0, 1.0, 0.0], [-sy, 0.0, cy]])
        rot_z = np.array([[cz, -sz, 0.0], [sz, cz, 0.0], [0.0, 0.0, 1.0]])
        rot = np.dot(rot_z, np.dot(rot_y, rot_x))
    out = np.dot(stretch, np.dot(rot, X.T)).T
    out += ctr
    return out


def anisotropy_matrices(ndim, scaling, angle):
    """(rot, stretch_diagonal) of adjust_for_anisotropy -- the same NumPy calls, so that the device-side generation of grid
    points (mik_set_grid) multiplies by bit-identical matrix entries."""
    ang = np.asarray(angle, dtype=np.float64) * np.pi / 180
    if ndim == 2:
        c, s = np.cos(-ang[0]), np.sin(-ang[0])
        return np.array([[c, -s], [s, c]]), np.array([1.0, scaling[0]], dtype=np.float64)
    cx, sx = np.cos(-ang[0]), np.sin(-ang[0])
    cy, sy = np.cos(-ang[1]), np.sin(-ang[1])
    cz, sz = np.cos(-ang[2]), np.sin(-ang[2])
    rot_x = np.array([[1.0, 0.0, 0.0], [0.0, cx, -sx], [0.0, sx, cx]])
    rot_y = np.array([[cy, 0.0, sy], [0.0, 1.0, 0.0], [-sy, 0.0, cy]])
    rot_z = np.array([[cz, -sz, 0.0], [sz, cz, 0.0], [0.0, 0.0, 1.0]])
    return np.dot(rot_z, np.dot(rot_y, rot_x)), np.array([1.0, scaling[0], scaling[1]], dtype=np.float64)


def great_circle_distance(lon1, lat1, lon2, lat2):
    """Great-circle distance in degrees, arctan form (core.py:36-97) -- host twin of the device functor
    gc_dist; used by the constructor-time variogram fit only."""
    lat1 = np.array(lat1) * np.pi / 180.0
    lat2 = np.array(lat2) * np.pi / 180.0
    dlon = (lon1 - lon2) * np.pi / 180.0
    c1, s1, c2, s2, cd = np.cos(lat1), np.sin(lat1), np.cos(lat2), np.sin(lat2), np.cos(dlon)
    return 180.0 / np.pi * np.arctan2(np.sqrt((c2 * np.sin(dlon)) ** 2 + (c1 * s2 - s1 * c2 * cd) ** 2),
                                      s1 * s2 + c1 * c2 * cd)


def euclid3_to_great_circle(euclid3_distance):
    """Chord length between two points of the unit sphere -> their great-circle distance in degrees (core.py:100-117): what turns
    the KD-tree's 3-D distances of a geographic moving window back into the distances the variogram takes."""
    # chords that round above the diameter are the diameter (core.py:115-116: "eliminate some possible numerical errors")
    d = np.minimum(np.asarray(euclid3_distance, dtype=np.float64), 2.0)
    return 180.0 - 360.0 / np.pi * np.arccos(0.5 * d)


def make_variogram_parameter_list(model, params):
    """User parameters (list or dict) -> internal list; None stays None (= 'fit it')."""
    if params is None:
        if model == "custom":  # core.py:525-529
            raise ValueError("Variogram parameters must be specified when implementing custom variogram model.")
        return None
    if model == "custom":  # core.py:309-314, 359-360: a list, handed to the user's function as is
        if type(params) is dict:
            raise TypeError("For user-specified custom variogram model, parameters must be specified in a list, not a dict.")
        if type(params) is list:
            return list(params)
        raise TypeError("Variogram model parameters must be provided in either a list or a dict")
    if model not in MODELS:
        raise ValueError("Specified variogram model must be one of the following: " + ", ".join(repr(m) for m in MODELS))
    if type(params) is dict:
        need = {"linear": ("slope", "nugget"), "power": ("scale", "exponent", "nugget")}.get(model, ("range", "nugget"))
        missing = [k for k in need if k not in params]
        if missing:
            raise KeyError("'%s' variogram model requires %s in the variogram parameter dictionary" % (model, need))
        if model == "linear":
            return [params["slope"], params["nugget"]]
        if model == "power":
            return [params["scale"], params["exponent"], params["nugget"]]
        if "sill" in params:
            return [params["sill"] - params["nugget"], params["range"], params["nugget"]]
        if "psill" in params:
            return [params["psill"], params["range"], params["nugget"]]
        raise KeyError("'%s' variogram model requires either 'sill' or 'psill'" % model)
    if type(params) is list:
        want = 2 if model == "linear" else 3
        if len(params) != want:
            raise ValueError("Variogram model parameter list must have exactly %d entries for '%s'" % (want, model))
        if model in BOUNDED:  # the list form carries the FULL sill
            return [params[0] - params[2], params[1], params[2]]
        return list(params)
    raise TypeError("Variogram model parameters must be provided in either a list or a dict")


def variogram_value(model, m, d):
    """Variogram models on the host (variogram_models.py:25-81) -- used only by the constructor-time
    fit, never by execute()."""
    d = np.asarray(d, dtype=np.float64)
    if model == "linear":
        return float(m[0]) * d + float(m[1])
    if model == "power":
        return float(m[0]) * d ** float(m[1]) + float(m[2])
    psill, rng, nug = float(m[0]), float(m[1]), float(m[2])
    if model == "gaussian":
        return psill * (1.0 - np.exp(-(d**2.0) / (rng * 4.0 / 7.0) ** 2.0)) + nug
    if model == "exponential":
        return psill * (1.0 - np.exp(-d / (rng / 3.0))) + nug
    if model == "spherical":
        return np.where(d <= rng, psill * ((3.0 * d) / (2.0 * rng) - (d**3.0) / (2.0 * rng**3.0)) + nug, psill + nug)
    if model == "hole-effect":
        return psill * (1.0 - (1.0 - d / (rng / 3.0)) * np.exp(-d / (rng / 3.0))) + nug
    raise ValueError("unknown variogram model %r" % (model,))


def bilinear_zscalars(zgrid, gx, gy, x, y):
    """external_Z drift: bilinear interpolation of zgrid (ny, nx) on axes gx, gy at ORIGINAL (unadjusted)
    coordinates x, y (any shape).  Raises ValueError outside the grid, like the reference."""
    gx = np.asarray(gx, dtype=np.float64).ravel()
    gy = np.asarray(gy, dtype=np.float64).ravel()
    zgrid = np.asarray(zgrid, dtype=np.float64)
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    if x.size and (x.max() > gx.max() or x.min() < gx.min() or y.max() > gy.max() or y.min() < gy.min()):
        raise ValueError("External drift array does not cover specified kriging domain.")
    # The reference brackets with x2 = min{i: g_i >= v} and x1 = max{i: g_i <= v} over the axis AS GIVEN (uk.py:556-559).  On an
    # ascending axis those are the two neighbours; on a descending or unsorted axis (a north-up raster) they are whatever
    # indices the rule yields -- usually far-apart nodes -- and the reference interpolates between those.  Same rule here.
    def bracket(g, v):
        if g.size < 2 or np.all(np.diff(g) > 0.0):
            return np.searchsorted(g, v, side="right") - 1, np.searchsorted(g, v, side="left")
        i1, i2 = np.empty(v.size, dtype=np.intp), np.empty(v.size, dtype=np.intp)
        step = max(1, (1 << 22) // g.size)
        for s0 in range(0, v.size, step):
            vv = v[s0:s0 + step, None]
            i2[s0:s0 + step] = np.argmax(g[None, :] >= vv, axis=1)
            i1[s0:s0 + step] = g.size - 1 - np.argmax(g[None, ::-1] <= vv, axis=1)
        return i1, i2

    xf, yf = x.ravel(), y.ravel()
    ix1, ix2 = bracket(gx, xf)
    iy1, iy2 = bracket(gy, yf)
    sx, sy = gx, gy
    x1, x2, y1, y2 = sx[ix1], sx[ix2], sy[iy1], sy[iy2]
    jx1, jx2, jy1, jy2 = ix1, ix2, iy1, iy2
    z11, z12 = zgrid[jy1, jx1], zgrid[jy1, jx2]
    z21, z22 = zgrid[jy2, jx1], zgrid[jy2, jx2]
    same_x, same_y = ix1 == ix2, iy1 == iy2
    with np.errstate(divide="ignore", invalid="ignore"):
        both = (z11 * (x2 - xf) * (y2 - yf) + z12 * (xf - x1) * (y2 - yf) + z21 * (x2 - xf) * (yf - y1)
                + z22 * (xf - x1) * (yf - y1)) / ((x2 - x1) * (y2 - y1))
        only_x = (z11 * (x2 - xf) + z22 * (xf - x1)) / (x2 - x1)   # same y row (reference uses [y2,x2] == [y1,x2])
        only_y = (z11 * (y2 - yf) + z22 * (yf - y1)) / (y2 - y1)
    out = np.where(same_y, np.where(same_x, z11, only_x), np.where(same_x, only_y, both))
    return out.reshape(x.shape)


# ----------------------------------------------------------------------------------------------------------
# Device-backed twins of core._krige / core._find_statistics (core.py:654-836).  Same signatures; the variogram
# is identified by the function's __name__ (as lib/variogram_models.pyx:6-22 does), arbitrary callables are refused.
# ----------------------------------------------------------------------------------------------------------
_eps = 1.0e-10  # core.py:30


def _model_of(variogram_function):
    from . import variogram_models as vm

    name = getattr(variogram_function, "__name__", None)
    return vm.MODEL_OF_FUNCTION.get(name, "custom")  # anything else is a user callable: evaluated on the host


def _set_problem_on(h, X, y, variogram_function, variogram_model_parameters, coordinates_type, pseudo_inv=False):
    from . import _lib

    model = _model_of(variogram_function)
    params = [float(v) for v in variogram_model_parameters]
    if model == "custom":
        fn, par = variogram_function, variogram_model_parameters
        h.set_custom_variogram(lambda d: fn(par, d))
        params = [0.0, 0.0, 0.0]
    h.set_problem(ndim=X.shape[1], xs=X[:, 0], ys=X[:, 1], zs=X[:, 2] if X.shape[1] == 3 else None, values=y,
                  model_id=_lib.MODEL_IDS[model], params=params, eps=_eps,
                  exact_values=True, geographic=coordinates_type == "geographic", pseudo_inv=1 if pseudo_inv else 0)


def _problem_handle(X, y, variogram_function, variogram_model_parameters, coordinates_type, pseudo_inv=False):
    from . import _lib

    X = np.ascontiguousarray(X, dtype=np.float64)
    if coordinates_type not in ("euclidean", "geographic"):
        raise ValueError("Specified coordinate type '%s' is not supported." % coordinates_type)
    if coordinates_type == "geographic" and X.shape[1] != 2:
        raise ValueError("Geographic coordinate type only supported for 2D datasets.")
    h = _lib.Handle()
    _set_problem_on(h, X, y, variogram_function, variogram_model_parameters, coordinates_type, pseudo_inv)
    return h


def _krige(X, y, coords, variogram_function, variogram_model_parameters, coordinates_type, pseudo_inv=False):
    """Ordinary kriging of ONE point from the stations X (the single-point form of the execute() solve)."""
    coords = np.asarray(coords, dtype=np.float64).ravel()
    # pseudo_inv: the reference solves with numpy.linalg.lstsq (core.py:749-750), i.e. the minimum-norm solution pinv(A) b
    h = _problem_handle(X, y, variogram_function, variogram_model_parameters, coordinates_type, pseudo_inv)
    try:
        h.factor()
        h.set_points(coords[0:1], coords[1:2], coords[2:3] if coords.size == 3 else None)
        h.predict()
        z, ss = h.get_results()
    finally:
        h.close()
    return float(z[0]), float(ss[0])


def _statistics_pseudo_inv(h, set_subset, X, y, gamma, geographic):
    """Statistics with pseudo_inv=True: the reference solves each of the N-1 growing (singular, once a duplicated station has
    entered) systems with numpy.linalg.lstsq (core.py:749-750), i.e. x = pinv(A_i) b_i.  There is no recursion for that --
    the bordering identity of mik_statistics needs non-singular leading systems -- so station i is kriged from stations
    0..i-1 through the device pseudo-inverse (mik_problem.pseudo_inv: one-sided Jacobi, O(i^3) per station, O(N^4) in all,
    like the reference).  A completeness path for the few-hundred-station data sets that carry duplicates.

    `set_subset(i)` sets the problem on stations 0..i-1 (X[:i], adjusted coordinates); `gamma(d)` is the variogram on the host.
    core.py:728-730 zeroes b only at the FIRST station that coincides with the kriged one; the device's eps rule zeroes every
    coincident station (ok.py:665-672).  The two differ only when station i has two or more earlier duplicates and the nugget
    is non-zero: then the pseudo-inverse comes back from the device and the two dot products are formed here."""
    n = y.size
    k, ss = np.zeros(n), np.zeros(n)
    for i in range(1, n):
        set_subset(i)
        h.factor()
        if geographic:
            d = great_circle_distance(X[:i, 0], X[:i, 1], X[i, 0] * np.ones(i), X[i, 1] * np.ones(i))
        else:
            d = np.sqrt(((X[:i] - X[i]) ** 2).sum(axis=1))
        hits = np.flatnonzero(np.absolute(d) <= 1e-10)
        if hits.size >= 2:
            b = np.append(-np.asarray(gamma(d), dtype=np.float64), 1.0)
            b[hits[0]] = 0.0
            x = h.get_matrix(1) @ b
            k[i], ss[i] = x[:i] @ y[:i], -(x @ b)
            continue
        q = X[i]
        h.set_points(q[0:1], q[1:2], q[2:3] if q.size == 3 else None)
        h.predict()
        zi, si = h.get_results()
        k[i], ss[i] = zi[0], si[0]
    return k, ss


def _delta_sigma(y, k, ss, eps):
    delta, sigma = np.zeros(y.shape), np.zeros(y.shape)
    keep = np.absolute(ss) >= eps
    keep[0] = False
    with np.errstate(invalid="ignore"):
        delta[keep] = y[keep] - k[keep]
        sigma[keep] = np.sqrt(ss[keep])
    sel = sigma > eps
    return delta[sel], sigma[sel]


def _find_statistics(X, y, variogram_function, variogram_model_parameters, coordinates_type, pseudo_inv=False):
    """delta, sigma, epsilon of the variogram fit (station i kriged from stations 0..i-1) on the device."""
    y = np.ascontiguousarray(y, dtype=np.float64)
    X = np.ascontiguousarray(X, dtype=np.float64)
    if pseudo_inv:
        h = _problem_handle(X[:1], y[:1], variogram_function, variogram_model_parameters, coordinates_type, True)

        def subset(i):
            _set_problem_on(h, X[:i], y[:i], variogram_function, variogram_model_parameters, coordinates_type, True)

        try:
            k, ss = _statistics_pseudo_inv(h, subset, X, y, lambda d: variogram_function(variogram_model_parameters, d),
                                           coordinates_type == "geographic")
        finally:
            h.close()
    else:
        h = _problem_handle(X, y, variogram_function, variogram_model_parameters, coordinates_type)
        try:
            k, ss = h.statistics(y.size)
        finally:
            h.close()
    delta, sigma = _delta_sigma(y, k, ss, _eps)
    return delta, sigma, delta / sigma


def calcQ1(epsilon):
    return abs(np.sum(epsilon) / (epsilon.shape[0] - 1))


def calcQ2(epsilon):
    return np.sum(epsilon**2) / (epsilon.shape[0] - 1)


def calc_cR(Q2, sigma):
    return Q2 * np.exp(np.sum(np.log(sigma**2)) / sigma.shape[0])


eps = 1.0e-10  # core.py:30: the reference's cut-off for "is this distance zero"


def _scipy_pinv(kind):
    def f(a):
        import scipy.linalg as spl

        return getattr(spl, kind)(a)

    f.__name__ = kind
    return f


# core.py:33: the pseudo-inverse routines `pseudo_inv_type` names, as host callables for code that indexes the table itself;
# the classes run the pseudo-inverse on the device (mik_problem.pseudo_inv)
P_INV = {"pinv": _scipy_pinv("pinv"), "pinvh": _scipy_pinv("pinvh")}


# ---------------------------------------------------------------------------------------------------------------------
# The reference's private names for the host helpers above and for the constructor-time variogram estimation (core.py:120, 196,
# 379, 538, 582), same arguments and return values, so that code written against pykrige.core keeps running.
# ---------------------------------------------------------------------------------------------------------------------
_adjust_for_anisotropy = adjust_for_anisotropy
_make_variogram_parameter_list = make_variogram_parameter_list


def _variogram_residuals(params, x, y, variogram_function, weight):
    """core.py:538-579: residuals of variogram_function(params, x) against y, optionally lag-weighted."""
    from . import variogram_fit

    return variogram_fit.residuals(params, x, y, variogram_function, weight)


def _calculate_variogram_model(lags, semivariance, variogram_model, variogram_function, weight):
    """core.py:582-651: the fitted parameters (internal order: slope, nugget / scale, exponent, nugget / psill, range, nugget)."""
    from . import variogram_fit

    return variogram_fit.calculate(lags, semivariance, variogram_model, variogram_function, weight)


def _initialize_variogram_model(X, y, variogram_model, variogram_model_parameters, variogram_function, nlags, weight,
                                coordinates_type):
    """core.py:379-535: experimental semivariogram of (X, y) in nlags equal-width bins, and the model parameters -- the caller's
    (their number checked) or fitted.  Returns (lags, semivariance, variogram_model_parameters)."""
    from . import variogram_fit

    X, y = np.asarray(X, dtype=np.float64), np.asarray(y, dtype=np.float64)
    if coordinates_type == "geographic":
        if X.shape[1] != 2:
            raise ValueError("Geographic coordinate type only supported for 2D datasets.")
    elif coordinates_type != "euclidean":
        raise ValueError("Specified coordinate type '%s' is not supported." % coordinates_type)
    lags, semivariance = variogram_fit.experimental_variogram(X, y, nlags, coordinates_type)
    if variogram_model_parameters is not None:
        if variogram_model == "linear" and len(variogram_model_parameters) != 2:
            raise ValueError("Exactly two parameters required for linear variogram model.")
        if variogram_model in ("power", "spherical", "exponential", "gaussian", "hole-effect") and len(variogram_model_parameters) != 3:
            raise ValueError("Exactly three parameters required for %s variogram model" % variogram_model)
        return lags, semivariance, variogram_model_parameters
    if variogram_model == "custom":
        raise ValueError("Variogram parameters must be specified when implementing custom variogram model.")
    return lags, semivariance, _calculate_variogram_model(lags, semivariance, variogram_model, variogram_function, weight)
