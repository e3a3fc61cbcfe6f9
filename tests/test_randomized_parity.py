"""Seeded randomized differential test: random kriging set-ups (dimension, size, variogram, anisotropy, drift terms,
style, mask, backend, moving window) through the drop-in classes on the GPU against the CPU oracle.  Every case is
reproducible from its seed (printed in the assertion message)."""
import numpy as np
import pytest

from oracle import kriging_oracle as ko

pytestmark = pytest.mark.gpu

Z_TOL, SS_TOL = 1e-8, 1e-6
MODELS = ["linear", "power", "gaussian", "spherical", "exponential", "hole-effect"]


def _case(seed):
    rng = np.random.default_rng(seed)
    ndim = int(rng.choice([2, 2, 3]))
    n = int(rng.choice([3, 5, 17, 64, 129, 200, 300, 450]))
    model = str(rng.choice(MODELS))
    if model == "linear":
        user = [float(rng.uniform(0.5, 2.0)), float(rng.uniform(0.0, 0.2))]
    elif model == "power":
        user = [float(rng.uniform(0.5, 2.0)), float(rng.uniform(0.6, 1.6)), float(rng.uniform(0.0, 0.2))]
    else:  # [sill, range, nugget]; a nugget keeps the gaussian / hole-effect systems well conditioned
        user = [float(rng.uniform(0.8, 2.0)), float(rng.uniform(0.3, 1.2)), float(rng.uniform(0.02, 0.2))]
    coords = rng.random((n, ndim))
    values = np.sin(5 * coords[:, 0]) * np.cos(3 * coords[:, 1]) + 0.1 * rng.standard_normal(n)
    scaling = [float(rng.uniform(0.5, 3.0)) for _ in range(ndim - 1)] if rng.random() < 0.6 else [1.0] * (ndim - 1)
    angle = [float(rng.uniform(-90, 90)) for _ in range(2 * ndim - 3)] if rng.random() < 0.6 else [0.0] * (2 * ndim - 3)
    exact = bool(rng.random() < 0.7)
    universal = bool(rng.random() < 0.5) and n >= 8
    drift = {}
    if universal:
        drift["regional_linear"] = bool(rng.random() < 0.7)
        if ndim == 2 and rng.random() < 0.5:
            w = int(rng.integers(1, 4))
            drift["wells"] = np.column_stack([rng.random((w, 2)), rng.uniform(-2, 2, w)])
        if rng.random() < 0.4:
            drift["specified"] = [np.cos(4 * coords[:, 0]) + coords[:, 1]]
        if rng.random() < 0.4:
            drift["functional"] = True
    geographic = bool(ndim == 2 and not universal and rng.random() < 0.2)
    if geographic:  # lon / lat in degrees, great-circle distances, no anisotropy (ok.py:289-304)
        coords = np.column_stack([rng.uniform(-170, 170, n), rng.uniform(-80, 80, n)])
        values = np.sin(np.radians(coords[:, 0])) * np.cos(np.radians(coords[:, 1])) + 0.1 * rng.standard_normal(n)
        scaling, angle = [1.0], [0.0]
        if model not in ("linear", "power"):
            user[1] = float(rng.uniform(15.0, 70.0))  # range in degrees
        else:
            user[0] = float(rng.uniform(0.005, 0.05))
    style = str(rng.choice(["grid", "points", "masked"]))
    sizes = [int(rng.integers(2, 12)) for _ in range(ndim)]
    if style == "points":
        axes = [rng.uniform(-0.1, 1.1, sizes[0]) for _ in range(ndim)] if not geographic else \
            [rng.uniform(-175, 175, sizes[0]), rng.uniform(-85, 85, sizes[0])]
        k_hit = min(2, sizes[0], n)
        for d in range(ndim):  # a couple of points on stations: the eps rule
            axes[d][:k_hit] = coords[:k_hit, d]
        shape = (sizes[0],)
    else:
        axes = [np.linspace(0.0, 1.0, s) for s in sizes] if not geographic else \
            [np.linspace(-160.0, 160.0, sizes[0]), np.linspace(-75.0, 75.0, sizes[1])]
        shape = tuple(sizes[::-1])
    mask = (rng.random(shape) < 0.3) if style == "masked" else None
    window = None
    if not universal and n >= 12 and rng.random() < 0.5:
        window = int(rng.integers(2, min(n, 40)))
    backend = str(rng.choice(["loop", "C"] if (ndim == 2 and not universal) else ["loop"])) if window else \
        str(rng.choice(["vectorized", "loop", "hip"] + (["C"] if (ndim == 2 and not universal) else [])))
    return dict(seed=seed, ndim=ndim, n=n, model=model, user=user, coords=coords, values=values, scaling=scaling, angle=angle,
                exact=exact, universal=universal, drift=drift, style=style, axes=axes, shape=shape, mask=mask, window=window,
                backend=backend, geographic=geographic)


def _functional_terms(ndim):
    if ndim == 2:
        return [lambda x, y: x * y, lambda x, y: np.sin(2 * x)]
    return [lambda x, y, z: x * z, lambda x, y, z: np.cos(y)]


def _spec_at_points(c, pts):
    return [np.cos(4 * pts[0]) + pts[1]]


_SEEN = []  # (seed, cond, |dz|, |dss|, z bar widened?, ss bar widened?, tag) of every case of this run


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("MIK_FUZZ_CASES", "120"))))
def test_random_configuration_against_the_oracle(seed):
    import pykrige_amd as pa

    c = _case(seed)
    ndim, coords, values = c["ndim"], c["coords"], c["values"]
    fun = _functional_terms(ndim) if c["drift"].get("functional") else []
    spec_st = c["drift"].get("specified", [])
    st = ko.KrigingState(ndim=ndim, coords_orig=coords, values=values, model=c["model"],
                         params=ko.internal_parameters(c["model"], c["user"]), scaling=c["scaling"], angle=c["angle"],
                         exact_values=c["exact"], regional_linear=bool(c["drift"].get("regional_linear")),
                         point_log=c["drift"].get("wells"), specified_data=list(spec_st), functional=list(fun),
                         geographic=c["geographic"])
    if c["style"] == "points":
        spec_pts = _spec_at_points(c, c["axes"]) if spec_st else []
    else:
        grids = np.meshgrid(*c["axes"]) if ndim == 2 else np.meshgrid(c["axes"][2], c["axes"][1], c["axes"][0], indexing="ij")[::-1]
        spec_pts = [np.cos(4 * grids[0]) + grids[1]] if spec_st else []
    kw = dict(variogram_model=c["model"], variogram_parameters=list(c["user"]), exact_values=c["exact"])
    if ndim == 2:
        kw.update(anisotropy_scaling=c["scaling"][0], anisotropy_angle=c["angle"][0])
        if c["geographic"]:
            kw["coordinates_type"] = "geographic"
        args = (coords[:, 0], coords[:, 1], values)
    else:
        kw.update(anisotropy_scaling_y=c["scaling"][0], anisotropy_scaling_z=c["scaling"][1], anisotropy_angle_x=c["angle"][0],
                  anisotropy_angle_y=c["angle"][1], anisotropy_angle_z=c["angle"][2])
        args = (coords[:, 0], coords[:, 1], coords[:, 2], values)
    if c["universal"]:
        terms = []
        if c["drift"].get("regional_linear"):
            terms.append("regional_linear")
        if "wells" in c["drift"]:
            terms.append("point_log")
            kw["point_drift"] = c["drift"]["wells"]
        if spec_st:
            terms.append("specified")
            kw["specified_drift"] = spec_st
        if fun:
            terms.append("functional")
            kw["functional_drift"] = fun
        kw["drift_terms"] = terms
        m = (pa.UniversalKriging if ndim == 2 else pa.UniversalKriging3D)(*args, **kw)
    else:
        m = (pa.OrdinaryKriging if ndim == 2 else pa.OrdinaryKriging3D)(*args, **kw)
    ekw = dict(backend=c["backend"])
    if c["mask"] is not None:
        ekw["mask"] = c["mask"]
    if spec_pts:
        ekw["specified_drift_arrays"] = spec_pts
    if c["window"]:
        ekw["n_closest_points"] = c["window"]
    z, ss = m.execute(c["style"], *c["axes"], **ekw)
    if c["window"]:
        pts = np.stack([g.ravel() for g in (np.meshgrid(*c["axes"]) if ndim == 2 else
                                            np.meshgrid(c["axes"][2], c["axes"][1], c["axes"][0], indexing="ij")[::-1])], 1) \
            if c["style"] != "points" else np.stack(c["axes"], 1)
        pts_adj = pts.copy() if c["geographic"] else ko.adjust_for_anisotropy(pts.copy(), st.center, st.scaling, st.angle)
        zr, sr = ko.solve_points_moving_window(st, pts_adj, c["window"])
        zr, sr = zr.reshape(c["shape"]), sr.reshape(c["shape"])
    else:
        zr, sr = ko.execute(st, c["style"], *c["axes"], mask=c["mask"], specified_drift_arrays=spec_pts)
    tag = "seed %d: %dD%s n=%d %s %s %s window=%s backend=%s drift=%s" % (
        seed, ndim, " geographic" if c["geographic"] else "", c["n"], c["model"], c["style"], "UK" if c["universal"] else "OK",
        c["window"], c["backend"], sorted(c["drift"]))
    assert z.shape == c["shape"] and ss.shape == c["shape"], tag
    keep = np.ones(c["shape"], bool) if c["mask"] is None else ~c["mask"]
    cond = np.linalg.cond(ko.kriging_matrix(st)) if not c["window"] else 1.0
    ztol = max(Z_TOL, 1e-15 * cond)   # an ill-conditioned matrix moves the reference's own LAPACK answer by cond * eps too
    stol = max(SS_TOL, 1e-15 * cond)
    dz = float(np.abs(np.ma.getdata(z)[keep] - np.ma.getdata(zr)[keep]).max()) if keep.any() else 0.0
    ds = float(np.abs(np.ma.getdata(ss)[keep] - np.ma.getdata(sr)[keep]).max()) if keep.any() else 0.0
    _SEEN.append((seed, cond, dz, ds, ztol > Z_TOL, stol > SS_TOL, tag))
    np.testing.assert_allclose(np.ma.getdata(z)[keep], np.ma.getdata(zr)[keep], rtol=0, atol=ztol, err_msg=tag)
    np.testing.assert_allclose(np.ma.getdata(ss)[keep], np.ma.getdata(sr)[keep], rtol=0, atol=stol, err_msg=tag)
    if c["backend"] == "vectorized":
        assert isinstance(z, np.ma.MaskedArray), tag
    elif c["style"] != "masked":
        assert type(z) is np.ndarray, tag


def test_zz_how_often_the_widened_bar_was_needed():
    """The bar is max(1e-8, cond(A) 1e-15) on z and max(1e-6, cond(A) 1e-15) on sigma^2.  This prints how often the second
    term was the larger one, and whether any of those cases actually NEEDED it (error above the plain BASELINE bar)."""
    if not _SEEN:
        pytest.skip("runs after the randomized cases")
    wide_z = [c for c in _SEEN if c[4]]
    wide_s = [c for c in _SEEN if c[5]]
    need_z = [c for c in wide_z if c[2] > Z_TOL]
    need_s = [c for c in wide_s if c[3] > SS_TOL]
    print("\nrandomized parity: %d cases; bar widened by cond(A) on z in %d (needed in %d), on sigma^2 in %d (needed in %d); "
          "worst |dz| %.2e, worst |dss| %.2e" % (len(_SEEN), len(wide_z), len(need_z), len(wide_s), len(need_s),
                                                 max(c[2] for c in _SEEN), max(c[3] for c in _SEEN)))
    for c in need_z + need_s:
        print("  needed the widened bar: cond %.2e |dz| %.2e |dss| %.2e  %s" % (c[1], c[2], c[3], c[6]))
