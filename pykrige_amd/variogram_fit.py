"""Constructor-time variogram estimation (host, SciPy) -- NOT on the execute() path.

Used only when a kriging object is built without explicit variogram parameters.  Restates the
behaviour of the reference's _initialize_variogram_model / _calculate_variogram_model
(/root/reference/src/pykrige/core.py:379-651): equal-width lag bins over the pairwise distances,
bin means of distance and semivariance, then a bounded soft-L1 least-squares fit of the model.
SURVEY.md section 8(f) rank 4 lists moving the O(N^2) pair reduction to the GPU as a later step.
"""
import numpy as np
from scipy.optimize import least_squares
from scipy.spatial.distance import pdist

from . import core


def experimental_variogram(coords, values, nlags, coordinates_type="euclidean"):
    values = np.asarray(values, dtype=np.float64)
    if coordinates_type == "geographic":  # core.py:437-451: all pairs i > j of the great-circle distance matrix
        x1, x2 = np.meshgrid(coords[:, 0], coords[:, 0], sparse=True)
        y1, y2 = np.meshgrid(coords[:, 1], coords[:, 1], sparse=True)
        z1, z2 = np.meshgrid(values, values, sparse=True)
        dm = core.great_circle_distance(x1, y1, x2, y2)
        gm = 0.5 * (z1 - z2) ** 2.0
        lower = np.tril_indices(dm.shape[0], -1)
        d, g = dm[lower], gm[lower]
    else:
        d = pdist(coords, metric="euclidean")
        g = 0.5 * pdist(values[:, None], metric="sqeuclidean")
    dmin, dmax = np.amin(d), np.amax(d)
    width = (dmax - dmin) / nlags
    edges = [dmin + k * width for k in range(nlags)] + [dmax + 0.001]
    lags, semis = [], []
    for k in range(nlags):
        sel = (d >= edges[k]) & (d < edges[k + 1])
        if sel.any():
            lags.append(np.mean(d[sel]))
            semis.append(np.mean(g[sel]))
    return np.array(lags), np.array(semis)


def residuals(params, lags, semis, gamma, weight):
    """Misfit of the model curve gamma(params, lags) against the binned semivariances (core.py:538-579 _variogram_residuals);
    weight: logistic weights centred at 70 % of the lag range, normalised to sum 1."""
    r = gamma(params, lags) - semis
    if weight:
        span = np.amax(lags) - np.amin(lags)
        k = 2.1972 / (0.1 * span)
        x0 = 0.7 * span + np.amin(lags)
        w = 1.0 / (1.0 + np.exp(-k * (x0 - lags)))
        r = r * (w / np.sum(w))
    return r


def _residuals(params, lags, semis, model, weight):
    return residuals(params, lags, semis, lambda p, d: core.variogram_value(model, p, d), weight)


def calculate(lags, semis, model, gamma=None, weight=False):
    """Bounded soft-L1 least-squares fit of the model to (lags, semis) -> parameters in the internal order
    (core.py:582-651 _calculate_variogram_model); gamma(params, lags) = the model curve (None: the named model's)."""
    lags, semis = np.asarray(lags, dtype=np.float64), np.asarray(semis, dtype=np.float64)
    if gamma is None:
        gamma = lambda p, d: core.variogram_value(model, p, d)  # noqa: E731
    smax, smin, lmax, lmin = np.amax(semis), np.amin(semis), np.amax(lags), np.amin(lags)
    if model == "linear":
        x0 = [(smax - smin) / (lmax - lmin), smin]
        bounds = ([0.0, 0.0], [np.inf, smax])
    elif model == "power":
        x0 = [(smax - smin) / (lmax - lmin), 1.1, smin]
        bounds = ([0.0, 0.001, 0.0], [np.inf, 1.999, smax])
    else:
        x0 = [smax - smin, 0.25 * lmax, smin]
        bounds = ([0.0, 0.0, 0.0], [10.0 * smax, lmax, smax])
    return least_squares(residuals, x0, bounds=bounds, loss="soft_l1", args=(lags, semis, gamma, weight)).x


def fit(coords, values, model, nlags=6, weight=False, coordinates_type="euclidean", binned=None):
    """binned = (lags, semivariance) already computed (on the device); None -> bin here on the host."""
    lags, semis = binned if binned is not None else experimental_variogram(coords, values, nlags, coordinates_type)
    return lags, semis, list(calculate(lags, semis, model, None, weight))
