"""Helpers shared by the oracle tests (CPU) and the HIP parity tests (GPU)."""
import glob
import os

import numpy as np

from oracle import kriging_oracle as ko

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# functional drifts cannot be stored in an .npz; they are re-declared here exactly as in oracle/make_golden.py
FUNCS = {
    "uk2d_spec_func": [lambda a, b: a * b, lambda a, b: a**2],
    "uk3d_rl_func": [lambda a, b, c: a * c],
    "uk3d": [lambda a, b, c: a * c],  # tests/golden/fullsize/uk3d.npz
}


NON_EXECUTE = ("fit_variograms", "pseudo_dup", "mw_ok2d", "mw_ok3d", "stats_find_statistics", "sk_callers", "tools_grid_files", "custom_variogram", "aniso_adjust", "r2_host_rules")  # fixtures that are not (stations, grid) -> (z, ss) cases


def names():
    return sorted(n for n in (os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN, "*.npz")))
                  if n not in NON_EXECUTE)


def external_z(g, x, y):
    """external_Z drift values by an implementation independent of the product's: SciPy's linear
    RegularGridInterpolator on the (demy, demx) grid = bilinear interpolation (uk.py:512-628)."""
    from scipy.interpolate import RegularGridInterpolator

    f = RegularGridInterpolator((g["demy"], g["demx"]), g["dem"], method="linear")
    return f(np.stack([np.ravel(y), np.ravel(x)], axis=1)).reshape(np.shape(x))


def load(name):
    with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as f:
        return {k: f[k] for k in f.files}


def state_from(name, g):
    """Oracle KrigingState for a golden fixture."""
    ndim = 3 if "zc" in g else 2
    coords = np.stack([g["x"], g["y"]] + ([g["zc"]] if ndim == 3 else []), axis=1)
    model = str(g["model"])
    scaling = np.atleast_1d(g["scaling"]).tolist() if "scaling" in g else [1.0] * (ndim - 1)
    angle = np.atleast_1d(g["angle"]).tolist() if "angle" in g else [0.0] * (2 * ndim - 3)
    rl = bool(g["regional_linear"]) if "regional_linear" in g else name in ("uk2d_spec_func", "uk3d_rl_func")
    return ko.KrigingState(
        ndim=ndim, coords_orig=coords, values=g["v"], model=model,
        params=ko.internal_parameters(model, g["params_user"].tolist()),
        scaling=scaling, angle=angle,
        exact_values=bool(g["exact"]) if "exact" in g else True,
        regional_linear=rl,
        point_log=g["wells"] if "wells" in g else None,
        specified_data=([external_z(g, g["x"], g["y"])] if "dem" in g else []) + ([g["spec_data"]] if "spec_data" in g else []),
        functional=FUNCS.get(name, []),
        geographic=bool(g["geographic"]) if "geographic" in g else False,
    )


def spec_point_arrays(g):
    """Per-point drift arrays the ORACLE needs (external_Z evaluated on the grid, then specified)."""
    out = []
    if "dem" in g:
        GX, GY = np.meshgrid(g["gridx"], g["gridy"])
        out.append(external_z(g, GX, GY))
    if "spec_grid" in g:
        out.append(g["spec_grid"])
    return out


def grid_args(g):
    if "gridz" in g:
        return (g["gridx"], g["gridy"], g["gridz"])
    return (g["gridx"], g["gridy"])


def amd_model_from(name, g):
    """pykrige_amd kriging object for a golden fixture (same constructor keywords as the reference)."""
    import pykrige_amd as pa

    ndim = 3 if "zc" in g else 2
    model = str(g["model"])
    params = [float(v) for v in g["params_user"].tolist()]
    exact = bool(g["exact"]) if "exact" in g else True
    rl = bool(g["regional_linear"]) if "regional_linear" in g else name in ("uk2d_spec_func", "uk3d_rl_func")
    terms = []
    kw = {}
    if rl:
        terms.append("regional_linear")
    if "wells" in g:
        terms.append("point_log")
        kw["point_drift"] = g["wells"]
    if "dem" in g:
        terms.append("external_Z")
        kw.update(external_drift=g["dem"], external_drift_x=g["demx"], external_drift_y=g["demy"])
    if "spec_data" in g:
        terms.append("specified")
        kw["specified_drift"] = [g["spec_data"]]
    if name in FUNCS:
        terms.append("functional")
        kw["functional_drift"] = FUNCS[name]
    if ndim == 2:
        aniso = dict(anisotropy_scaling=float(g["scaling"]) if "scaling" in g else 1.0,
                     anisotropy_angle=float(g["angle"]) if "angle" in g else 0.0)
        if terms:
            return pa.UniversalKriging(g["x"], g["y"], g["v"], variogram_model=model, variogram_parameters=params,
                                       drift_terms=terms, exact_values=exact, **aniso, **kw)
        if "geographic" in g and bool(g["geographic"]):
            return pa.OrdinaryKriging(g["x"], g["y"], g["v"], variogram_model=model, variogram_parameters=params,
                                      exact_values=exact, coordinates_type="geographic")
        return pa.OrdinaryKriging(g["x"], g["y"], g["v"], variogram_model=model, variogram_parameters=params,
                                  exact_values=exact, **aniso)
    sc = np.atleast_1d(g["scaling"]).tolist() if "scaling" in g else [1.0, 1.0]
    an = np.atleast_1d(g["angle"]).tolist() if "angle" in g else [0.0, 0.0, 0.0]
    aniso = dict(anisotropy_scaling_y=sc[0], anisotropy_scaling_z=sc[1], anisotropy_angle_x=an[0],
                 anisotropy_angle_y=an[1], anisotropy_angle_z=an[2])
    if terms:
        return pa.UniversalKriging3D(g["x"], g["y"], g["zc"], g["v"], variogram_model=model, variogram_parameters=params,
                                     drift_terms=terms, exact_values=exact, **aniso, **kw)
    return pa.OrdinaryKriging3D(g["x"], g["y"], g["zc"], g["v"], variogram_model=model, variogram_parameters=params,
                                exact_values=exact, **aniso)


def synth(seed, n, ndim):
    """SURVEY.md 8(d) synthetic stations: uniform in the unit square/cube, smooth field + noise."""
    rng = np.random.default_rng(seed)
    c = [rng.random(n) for _ in range(ndim)]
    v = np.sin(6 * c[0]) * np.cos(4 * c[1])
    if ndim == 3:
        v = v * np.cos(3 * c[2])
    v = v + 0.1 * rng.standard_normal(n)
    return c, v
