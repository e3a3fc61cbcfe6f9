"""CPU oracle for the PyKrige ``execute()`` hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT.

A from-scratch NumPy/SciPy restatement of what the reference computes between
``a = self._get_kriging_matrix(n)`` and ``return zvalues, sigmasq`` for the four kriging
classes.  It exists so the HIP path can be checked where ``/root/reference`` is absent
(the GPU box).  Parity of this oracle itself is PINNED: ``tests/test_oracle_golden.py``
checks it against ``tests/golden/*.npz`` produced by the real reference
(``oracle/make_golden.py``, PyKrige 1.7.3 imported from /root/reference/src).

Reference lines restated (all paths relative to /root/reference/src/pykrige):
  variogram models ............ variogram_models.py:25-81
  anisotropy transform ........ core.py:120-193
  kriging matrix (OK 2D/3D) ... ok.py:626-648, ok3d.py:603-622
  kriging matrix (UK 2D/3D) ... uk.py:861-920, uk3d.py:688-737
  RHS + solve + reductions .... ok.py:650-683, uk.py:922-1009, ok3d.py:624-657, uk3d.py:739-811
  grid/mask front matter ...... ok.py:842-900, uk.py:1163-1291, ok3d.py:827-898, uk3d.py:975-1121
Heavy arithmetic delegated to the same third-party calls the reference makes
(scipy.spatial.distance.cdist, scipy.linalg.inv, numpy.dot) -- numpy>=1.20, scipy>=1.5.4,<2
(pyproject.toml:72-75); versions used for pinning: numpy 2.2.6, scipy 1.15.3.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, List, Optional, Sequence

import numpy as np
import scipy.linalg
from scipy.spatial.distance import cdist

EPS = 1.0e-10  # ok.py:177, uk.py:210

MODEL_IDS = {"linear": 0, "power": 1, "gaussian": 2, "spherical": 3, "exponential": 4, "hole-effect": 5}


# --------------------------------------------------------------------------------------
# variogram models  (variogram_models.py:25-81) -- m is the *internal* parameter list
# --------------------------------------------------------------------------------------
def variogram(model: str, m: Sequence[float], d: np.ndarray) -> np.ndarray:
    d = np.asarray(d, dtype=np.float64)
    if model == "linear":  # :25-29  m=[slope, nugget]
        return float(m[0]) * d + float(m[1])
    if model == "power":  # :32-37  m=[scale, exponent, nugget]
        return float(m[0]) * d ** float(m[1]) + float(m[2])
    psill, rng, nugget = float(m[0]), float(m[1]), float(m[2])
    if model == "gaussian":  # :40-45
        return psill * (1.0 - np.exp(-(d**2.0) / (rng * 4.0 / 7.0) ** 2.0)) + nugget
    if model == "exponential":  # :48-53
        return psill * (1.0 - np.exp(-d / (rng / 3.0))) + nugget
    if model == "spherical":  # :56-70 (np.piecewise with d<=range)
        inside = psill * ((3.0 * d) / (2.0 * rng) - (d**3.0) / (2.0 * rng**3.0)) + nugget
        return np.where(d <= rng, inside, psill + nugget)
    if model == "hole-effect":  # :73-81
        return psill * (1.0 - (1.0 - d / (rng / 3.0)) * np.exp(-d / (rng / 3.0))) + nugget
    raise ValueError("unknown variogram model %r" % (model,))


def great_circle_distance(lon1, lat1, lon2, lat2):
    """core.py:36-97 (arctan form, degrees)."""
    lat1 = np.array(lat1) * np.pi / 180.0
    lat2 = np.array(lat2) * np.pi / 180.0
    dlon = (lon1 - lon2) * np.pi / 180.0
    c1, s1, c2, s2, cd = np.cos(lat1), np.sin(lat1), np.cos(lat2), np.sin(lat2), np.cos(dlon)
    return 180.0 / np.pi * np.arctan2(np.sqrt((c2 * np.sin(dlon)) ** 2 + (c1 * s2 - s1 * c2 * cd) ** 2),
                                      s1 * s2 + c1 * c2 * cd)


def internal_parameters(model: str, params: Sequence[float]) -> List[float]:
    """List-form user parameters -> internal list (core.py:330-357): for the four bounded
    models the user gives [sill, range, nugget] and the code stores [sill-nugget, range, nugget]."""
    params = [float(p) for p in params]
    if model in ("gaussian", "spherical", "exponential", "hole-effect"):
        return [params[0] - params[2], params[1], params[2]]
    return params


# --------------------------------------------------------------------------------------
# anisotropy  (core.py:120-193)
# --------------------------------------------------------------------------------------
def adjust_for_anisotropy(X, center, scaling, angle) -> np.ndarray:
    X = np.array(X, dtype=np.float64, copy=True)
    center = np.asarray(center, dtype=np.float64)[None, :]
    angle = np.asarray(angle, dtype=np.float64) * np.pi / 180
    X -= center
    nd = X.shape[1]
    if nd == 2:
        stretch = np.array([[1, 0], [0, scaling[0]]])
        c, s = np.cos(-angle[0]), np.sin(-angle[0])
        rot = np.array([[c, -s], [s, c]])
    elif nd == 3:
        stretch = np.array([[1.0, 0.0, 0.0], [0.0, scaling[0], 0.0], [0.0, 0.0, scaling[1]]])
        cx, sx = np.cos(-angle[0]), np.sin(-angle[0])
        cy, sy = np.cos(-angle[1]), np.sin(-angle[1])
        cz, sz = np.cos(-angle[2]), np.sin(-angle[2])
        rx = np.array([[1.0, 0.0, 0.0], [0.0, cx, -sx], [0.0, sx, cx]])
        ry = np.array([[cy, 0.0, sy], [0.0, 1.0, 0.0], [-sy, 0.0, cy]])
        rz = np.array([[cz, -sz, 0.0], [sz, cz, 0.0], [0.0, 0.0, 1.0]])
        rot = np.dot(rz, np.dot(ry, rx))
    else:
        raise ValueError("only 2D/3D supported")
    Xa = np.dot(stretch, np.dot(rot, X.T)).T
    Xa += center
    return Xa


# --------------------------------------------------------------------------------------
# problem description (what a fitted kriging object carries into execute())
# --------------------------------------------------------------------------------------
@dataclass
class KrigingState:
    """Everything ``execute`` reads from a constructed reference object."""

    ndim: int
    coords_orig: np.ndarray  # (n, ndim) columns x,y[,z] as given by the user
    values: np.ndarray  # (n,)
    model: str
    params: List[float]  # internal parameter list
    center: np.ndarray = None  # (ndim,)
    scaling: Sequence[float] = (1.0,)  # ndim-1 entries
    angle: Sequence[float] = (0.0,)  # 2*ndim-3 entries
    exact_values: bool = True
    # universal-kriging drift terms (reference order: regional_linear, point_log,
    # external_Z, specified, functional -- uk.py:877-910, uk3d.py:708-732)
    regional_linear: bool = False
    point_log: Optional[np.ndarray] = None  # (W,3) user x,y,strength (2D only)
    specified_data: List[np.ndarray] = field(default_factory=list)  # values at stations
    functional: List[Callable] = field(default_factory=list)
    geographic: bool = False  # coordinates_type='geographic' (2D OK only): lon/lat degrees, great-circle distances
    coords_adj: np.ndarray = None
    wells_adj: Optional[np.ndarray] = None

    def __post_init__(self):
        self.coords_orig = np.asarray(self.coords_orig, dtype=np.float64)
        self.values = np.asarray(self.values, dtype=np.float64)
        if self.center is None:  # ok.py:276-277
            self.center = (self.coords_orig.max(axis=0) + self.coords_orig.min(axis=0)) / 2.0
        if self.geographic:  # ok.py:289-304: no adjustment
            self.center = np.zeros(2)
            self.coords_adj = self.coords_orig.copy()
        else:
            self.coords_adj = adjust_for_anisotropy(self.coords_orig, self.center, self.scaling, self.angle)
        if self.point_log is not None:  # uk.py:461-470
            pl = np.atleast_2d(np.asarray(self.point_log, dtype=np.float64))
            self.wells_adj = np.zeros(pl.shape)
            self.wells_adj[:, 2] = pl[:, 2]
            self.wells_adj[:, :2] = adjust_for_anisotropy(pl[:, :2], self.center, self.scaling, self.angle)

    @property
    def n(self):
        return self.coords_adj.shape[0]

    @property
    def n_drift(self):
        k = 0
        if self.regional_linear:
            k += self.ndim
        if self.wells_adj is not None:
            k += self.wells_adj.shape[0]
        k += len(self.specified_data) + len(self.functional)
        return k


def _log_well(dx2dy2_sqrt, strength):
    """point_log drift value incl. the -inf -> -100 rule (uk.py:885-896, 957-966)."""
    with np.errstate(divide="ignore"):
        ld = np.log(dx2dy2_sqrt)
    ld = np.where(np.isinf(ld), -100.0, ld)
    return -strength * ld


def _drift_columns(st: KrigingState, pts_adj: np.ndarray, spec: Sequence[np.ndarray]) -> np.ndarray:
    """(npts, n_drift) drift values at arbitrary adjusted points, reference column order."""
    cols = []
    if st.regional_linear:
        for k in range(st.ndim):  # x, y[, z]  (uk.py:877-883; uk3d.py:708-717)
            cols.append(pts_adj[:, k])
    if st.wells_adj is not None:
        for w in range(st.wells_adj.shape[0]):
            r = np.sqrt((pts_adj[:, 0] - st.wells_adj[w, 0]) ** 2 + (pts_adj[:, 1] - st.wells_adj[w, 1]) ** 2)
            cols.append(_log_well(r, st.wells_adj[w, 2]))
    for arr in spec:
        cols.append(np.asarray(arr, dtype=np.float64).ravel())
    for f in st.functional:
        cols.append(f(*[pts_adj[:, k] for k in range(st.ndim)]))
    if not cols:
        return np.zeros((pts_adj.shape[0], 0))
    return np.stack(cols, axis=1)


def kriging_matrix(st: KrigingState) -> np.ndarray:
    """ok.py:626-648 / uk.py:861-920 / ok3d.py:603-622 / uk3d.py:688-737."""
    n, p = st.n, st.n_drift
    if st.geographic:  # ok.py:634-640
        d = great_circle_distance(st.coords_adj[:, 0][:, None], st.coords_adj[:, 1][:, None], st.coords_adj[:, 0],
                                  st.coords_adj[:, 1])
    else:
        d = cdist(st.coords_adj, st.coords_adj, "euclidean")
    a = np.zeros((n + p + 1, n + p + 1))
    a[:n, :n] = -variogram(st.model, st.params, d)
    np.fill_diagonal(a, 0.0)
    if p:
        f = _drift_columns(st, st.coords_adj, st.specified_data)
        a[:n, n : n + p] = f
        a[n : n + p, :n] = f.T
    a[n + p, :n] = 1.0
    a[:n, n + p] = 1.0
    a[n:, n:] = 0.0
    return a


def rhs(st: KrigingState, pts_adj: np.ndarray, spec_pts: Sequence[np.ndarray] = ()) -> np.ndarray:
    """(npt, M) right-hand sides (ok.py:665-673, uk.py:937-986)."""
    n, p = st.n, st.n_drift
    # 3D execute() feeds cdist columns in (z, y, x) order (ok3d.py:885-899, uk3d.py:1108-1122)
    rev = slice(None, None, -1) if st.ndim == 3 else slice(None)
    if st.geographic:  # ok.py:990-996
        bd = great_circle_distance(pts_adj[:, 0][:, None], pts_adj[:, 1][:, None], st.coords_adj[:, 0], st.coords_adj[:, 1])
    else:
        bd = cdist(pts_adj[:, rev], st.coords_adj[:, rev], "euclidean")
    b = np.zeros((pts_adj.shape[0], n + p + 1))
    b[:, :n] = -variogram(st.model, st.params, bd)
    if st.exact_values:
        b[:, :n][np.absolute(bd) <= EPS] = 0.0
    if p:
        b[:, n : n + p] = _drift_columns(st, pts_adj, spec_pts)
    b[:, n + p] = 1.0
    return b


def solve_points(st: KrigingState, pts_adj: np.ndarray, spec_pts: Sequence[np.ndarray] = (),
                 a_inv: Optional[np.ndarray] = None, chunk: int = 4096):
    """z and sigma^2 at adjusted points, 'vectorized' arithmetic (inverse, then dgemm)."""
    n = st.n
    if a_inv is None:
        a_inv = scipy.linalg.inv(kriging_matrix(st))
    npt = pts_adj.shape[0]
    z = np.zeros(npt)
    ss = np.zeros(npt)
    for lo in range(0, npt, chunk):
        hi = min(npt, lo + chunk)
        b = rhs(st, pts_adj[lo:hi], [np.asarray(s).ravel()[lo:hi] for s in spec_pts])
        x = np.dot(a_inv, b.T).T  # ok.py:679
        z[lo:hi] = np.sum(x[:, :n] * st.values, axis=1)  # ok.py:680
        ss[lo:hi] = np.sum(x * -b, axis=1)  # ok.py:681
    return z, ss


def solve_points_moving_window(st: KrigingState, pts_adj: np.ndarray, n_closest_points: int):
    """Moving-window ordinary kriging: cKDTree.query (ok.py:957-960; ok3d.py:901-904) then, per point,
    the (k+1)x(k+1) system cut out of the full kriging matrix (ok.py:722-758, ok3d.py:697-733)."""
    from scipy.spatial import cKDTree

    if st.n_drift:
        raise ValueError("moving window exists for ordinary kriging only")
    rev = slice(None, None, -1) if st.ndim == 3 else slice(None)
    if st.geographic:  # ok.py:930-970: KD-tree on unit-sphere Cartesian coordinates, great-circle distances afterwards
        def unit(ll):
            lo, la = ll[:, 0] * np.pi / 180.0, ll[:, 1] * np.pi / 180.0
            return np.stack([np.cos(lo) * np.cos(la), np.sin(lo) * np.cos(la), np.sin(la)], 1)

        _, bd_idx = cKDTree(unit(st.coords_adj)).query(unit(pts_adj), k=n_closest_points, eps=0.0)
        bd_all = great_circle_distance(pts_adj[:, 0][:, None], pts_adj[:, 1][:, None], st.coords_adj[bd_idx, 0],
                                       st.coords_adj[bd_idx, 1])
    else:
        tree = cKDTree(st.coords_adj[:, rev])
        bd_all, bd_idx = tree.query(pts_adj[:, rev], k=n_closest_points, eps=0.0)
    a_all = kriging_matrix(st)
    n = n_closest_points
    z, ss = np.zeros(pts_adj.shape[0]), np.zeros(pts_adj.shape[0])
    for i in range(pts_adj.shape[0]):
        sel = np.concatenate((bd_idx[i], [a_all.shape[0] - 1]))
        a = a_all[sel[:, None], sel]
        b = np.zeros(n + 1)
        b[:n] = -variogram(st.model, st.params, bd_all[i])
        if st.exact_values:
            b[:n][np.absolute(bd_all[i]) <= EPS] = 0.0
        b[n] = 1.0
        x = scipy.linalg.solve(a, b)
        z[i] = x[:n].dot(st.values[bd_idx[i]])
        ss[i] = -x.dot(b)
    return z, ss


def execute(st: KrigingState, style: str, xpoints, ypoints, zpoints=None, mask=None,
            specified_drift_arrays: Sequence[np.ndarray] = (), a_inv=None):
    """Front/back matter of the four ``execute`` methods + the solve.  Returns plain
    ndarrays for grid/points and MaskedArrays for style='masked' (the 'loop'/'C' rule)."""
    if style not in ("grid", "masked", "points"):
        raise ValueError("style argument must be 'grid', 'points', or 'masked'")
    axes = [np.atleast_1d(np.squeeze(np.array(v, copy=True, dtype=np.float64)))
            for v in ((xpoints, ypoints) if st.ndim == 2 else (xpoints, ypoints, zpoints))]
    sizes = [a.size for a in axes]
    spec_pts = list(specified_drift_arrays)
    if style in ("grid", "masked"):
        if st.ndim == 2:
            nx, ny = sizes
            shape = (ny, nx)
            gx, gy = np.meshgrid(axes[0], axes[1])  # ok.py:864
            pts = np.stack([gx.ravel(), gy.ravel()], axis=1)
        else:
            nx, ny, nz = sizes
            shape = (nz, ny, nx)
            gz, gy, gx = np.meshgrid(axes[2], axes[1], axes[0], indexing="ij")  # ok3d.py:863
            pts = np.stack([gx.ravel(), gy.ravel(), gz.ravel()], axis=1)
        if style == "masked":
            if mask is None:
                raise IOError("Must specify boolean masking array when style is 'masked'.")
            mask = np.asarray(mask)
            if mask.shape != shape:
                if st.ndim == 2 and mask.shape == shape[::-1]:
                    mask = mask.T
                elif st.ndim == 3 and mask.shape == shape[::-1]:
                    mask = mask.swapaxes(0, 2)
                else:
                    raise ValueError("Mask dimensions do not match specified grid dimensions.")
            mask = mask.ravel().astype(bool)
        fixed = []
        for s in spec_pts:  # uk.py:1231-1258
            s = np.asarray(s)
            if s.shape != shape and s.shape == shape[::-1]:
                s = s.T if st.ndim == 2 else s.swapaxes(0, 2)
            fixed.append(np.asarray(s, dtype=np.float64).ravel())
        spec_pts = fixed
    else:
        if len(set(sizes)) != 1:
            raise ValueError("xpoints and ypoints must have same dimensions")
        shape = (sizes[0],)
        pts = np.stack(axes, axis=1)
        spec_pts = [np.asarray(s, dtype=np.float64).ravel() for s in spec_pts]
    npt = pts.shape[0]
    pts_adj = pts if st.geographic else adjust_for_anisotropy(pts, st.center, st.scaling, st.angle)
    z = np.zeros(npt)
    ss = np.zeros(npt)
    if style == "masked":
        sel = np.nonzero(~mask)[0]
        zz, s2 = solve_points(st, pts_adj[sel], [s[sel] for s in spec_pts], a_inv)
        z[sel], ss[sel] = zz, s2
        z = np.ma.array(z, mask=mask)
        ss = np.ma.array(ss, mask=mask)
    else:
        z, ss = solve_points(st, pts_adj, spec_pts, a_inv)
    return z.reshape(shape), ss.reshape(shape)
