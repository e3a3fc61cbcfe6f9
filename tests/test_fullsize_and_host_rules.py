"""Round-2 parity additions.

* tests/golden/fullsize/c{2,3,4,5}.npz (+ uk3d.npz, round 3: UniversalKriging3D, N = 2000, regional_linear + functional drift,
  anisotropic) -- BASELINE configs 2-5 at their real station counts: one row slab of >= 16 384
  points of each config's own grid + 8 exact-hit nodes, kriged by the REAL reference's backend='vectorized' (config 2 also
  backend='C') -- oracle/make_golden_fullsize.py.  CPU: the oracle is pinned on a sub-sample; GPU: the HIP path on the whole
  slab at |dz| <= 1e-8, |dsigma^2| <= 1e-6, with cond_1(A) printed.
* tests/golden/r2_host_rules.npz -- host-side rules the advisor found deviating from the reference
  (oracle/make_golden_r2.py): external_Z look-up on descending / unsorted axes, update_variogram_model call sequences,
  statistics with pseudo_inv=True, and style='grid' ignoring a mask (ok.py:896).
"""
import os

import numpy as np
import pytest

from oracle import kriging_oracle as ko
from tests import _fixtures as fx

Z_TOL, SS_TOL = 1e-8, 1e-6
FULL = os.path.join(fx.GOLDEN, "fullsize")
CASES = ("c2", "c3", "c4", "c5", "uk3d")


def _full(name):
    with np.load(os.path.join(FULL, name + ".npz"), allow_pickle=False) as f:
        return {k: f[k] for k in f.files}


def _r2():
    return fx.load("r2_host_rules")


def _node_points(g):
    """Flat indices (in the slab's meshgrid order) of the 8 nodes that carry a station."""
    shape = (len(g["gridx"]), len(g["gridy"])) + ((len(g["gridz"]),) if "gridz" in g else ())
    idx = np.unravel_index(g["node_flat"], shape)  # (ix, iy[, iz])
    if len(shape) == 2:
        return idx[1] * shape[0] + idx[0]
    return (idx[2] * shape[1] + idx[1]) * shape[0] + idx[0]


# ------------------------------------------------------------------------------------------- CPU: oracle pinned at full size
@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_the_reference_at_full_station_count(name):
    g = _full(name)
    st = fx.state_from(name, g)
    assert st.n == {"c2": 5000, "c3": 2000, "c4": 4000, "c5": 8000, "uk3d": 2000}[name]
    assert g["z"].size >= 16384
    nodes = _node_points(g)
    sel = np.unique(np.concatenate([nodes, np.arange(0, g["z"].size, 37)[:600]]))
    axes = fx.grid_args(g)
    if st.ndim == 2:
        X, Y = np.meshgrid(*axes)
        pts = np.stack([X.ravel(), Y.ravel()], 1)[sel]
    else:
        Z, Y, X = np.meshgrid(axes[2], axes[1], axes[0], indexing="ij")
        pts = np.stack([X.ravel(), Y.ravel(), Z.ravel()], 1)[sel]
    z, ss = ko.solve_points(st, ko.adjust_for_anisotropy(pts, st.center, st.scaling, st.angle))
    # same LAPACK/BLAS calls on the same numbers; the slack is rounding times cond(A) ~ 1e6
    np.testing.assert_allclose(z, g["z"].ravel()[sel], rtol=0, atol=1e-9)
    np.testing.assert_allclose(ss, g["ss"].ravel()[sel], rtol=0, atol=1e-9)
    # exact interpolation at the station nodes (the eps rule at full size)
    np.testing.assert_allclose(g["z"].ravel()[nodes], g["v"][:8], rtol=0, atol=1e-9)
    if "z_c" in g:  # the reference's own C loop against its vectorized backend on the same slab
        assert np.abs(g["z_c"] - g["z"]).max() < 1e-9 and np.abs(g["ss_c"] - g["ss"]).max() < 1e-9


def test_oracle_moving_window_matches_the_reference_at_config2_size():
    """fullsize/mw_c2.npz (round 4): config 2's 5000 stations, a 17 000-point row slab, the reference's backend='C' with
    n_closest_points = 10 and 100 (cKDTree.query + lib/cok.pyx:98-193).  The oracle's window restatement on a sub-sample."""
    g = _full("mw_c2")
    assert g["x"].size == 5000 and g["z_k10"].size >= 16384
    st = ko.KrigingState(ndim=2, coords_orig=np.stack([g["x"], g["y"]], 1), values=g["v"], model=str(g["model"]),
                         params=ko.internal_parameters(str(g["model"]), g["params_user"].tolist()))
    X, Y = np.meshgrid(g["gridx"], g["gridy"])
    sel = np.arange(0, X.size, 53)[:300]
    pts = np.stack([X.ravel(), Y.ravel()], 1)[sel]
    for w in (10, 100):
        z, ss = ko.solve_points_moving_window(st, pts, w)
        np.testing.assert_allclose(z, g["z_k%d" % w].ravel()[sel], rtol=0, atol=1e-9)
        np.testing.assert_allclose(ss, g["ss_k%d" % w].ravel()[sel], rtol=0, atol=1e-9)


# ------------------------------------------------------------------------------------------- CPU: host rules
@pytest.mark.parametrize("variant", ["asc", "descy", "descxy", "perm"])
def test_external_z_lookup_follows_the_reference_index_rule(variant):
    from pykrige_amd import core

    g = _r2()
    ax, ay, dem = g["zs_%s_ax" % variant], g["zs_%s_ay" % variant], g["zs_%s_dem" % variant]
    got_p = core.bilinear_zscalars(dem, ax, ay, g["zs_qx"], g["zs_qy"])
    got_s = core.bilinear_zscalars(dem, ax, ay, g["zs_x"], g["zs_y"])
    np.testing.assert_allclose(got_p, g["zs_%s_points" % variant], rtol=0, atol=1e-13)
    np.testing.assert_allclose(got_s, g["zs_%s_stations" % variant], rtol=0, atol=1e-13)
    if variant != "asc":  # the rule really does give something else than nearest-neighbour interpolation there
        assert np.abs(g["zs_%s_points" % variant] - g["zs_asc_points"]).max() > 1e-3


def test_grid_style_ignores_a_mask_like_the_reference():
    import pykrige_amd as pa

    ok = pa.OrdinaryKriging([0.0, 1.0, 0.3], [0.0, 0.2, 0.9], [1.0, 2.0, 3.0], variogram_model="linear",
                            variogram_parameters=[1.0, 0.0])
    m = np.zeros((3, 4), dtype=bool)
    m[1, 2] = True
    for style_mask in (m, m.T, np.ones(5, dtype=bool)):  # right shape, transposed, nonsense: all ignored (ok.py:896)
        pts, shape, mask = ok._points_from("grid", (np.linspace(0, 1, 4), np.linspace(0, 1, 3)), style_mask)
        assert mask is None and shape == (3, 4) and pts.shape == (12, 2)
    pts, shape, mask = ok._points_from("masked", (np.linspace(0, 1, 4), np.linspace(0, 1, 3)), m.T)
    assert mask.shape == (12,) and mask.reshape(3, 4)[1, 2]


def test_update_variogram_model_signatures_are_the_references():
    import inspect

    import pykrige_amd as pa

    p2 = inspect.signature(pa.OrdinaryKriging.update_variogram_model).parameters
    assert list(p2)[1:] == ["variogram_model", "variogram_parameters", "variogram_function", "nlags", "weight",
                            "anisotropy_scaling", "anisotropy_angle"]  # ok.py:379-387
    assert p2["anisotropy_scaling"].default == 1.0 and p2["anisotropy_angle"].default == 0.0
    assert inspect.signature(pa.UniversalKriging.update_variogram_model) == inspect.signature(pa.OrdinaryKriging.update_variogram_model)
    p3 = inspect.signature(pa.OrdinaryKriging3D.update_variogram_model).parameters
    assert list(p3)[6:] == ["anisotropy_scaling_y", "anisotropy_scaling_z", "anisotropy_angle_x", "anisotropy_angle_y",
                            "anisotropy_angle_z"]  # ok3d.py:368-380
    assert [p3[k].default for k in list(p3)[6:]] == [1.0, 1.0, 0.0, 0.0, 0.0]
    # host side effect, no device needed: omitted anisotropy resets to isotropic and re-adjusts the stations
    g = _r2()
    ok = pa.OrdinaryKriging(g["upd_x"], g["upd_y"], g["upd_v"], variogram_model="exponential",
                            variogram_parameters=[1.0, 0.3, 0.0], anisotropy_scaling=3.0, anisotropy_angle=45.0)
    assert np.abs(ok.X_ADJUSTED - g["upd_x"]).max() > 1e-3
    ok.update_variogram_model("spherical", [1.0, 0.5, 0.05])
    assert ok.anisotropy_scaling == 1.0 and ok.anisotropy_angle == 0.0
    np.testing.assert_allclose(ok.X_ADJUSTED, g["upd_x"], rtol=0, atol=1e-15)
    uk = pa.UniversalKriging(g["upd_x"], g["upd_y"], g["upd_v"], variogram_model="exponential",
                             variogram_parameters=[1.0, 0.3, 0.01], drift_terms=["point_log"], point_drift=g["upd_wells"],
                             anisotropy_scaling=2.0, anisotropy_angle=30.0)
    wells_before = uk.point_log_array.copy()
    uk.update_variogram_model("exponential", [1.0, 0.3, 0.01])
    np.testing.assert_array_equal(uk.point_log_array, wells_before)  # uk.py:710-723 re-adjusts the stations only


# ------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["default", "full_sweep", "half_sweep", "pivoted"])
@pytest.mark.parametrize("name", CASES)
def test_hip_matches_the_reference_on_a_full_size_slab(name, variant):
    g = _full(name)
    m = fx.amd_model_from(name, g)
    h = m._get_handle()
    if variant == "half_sweep":  # inverse from the upper block triangle only
        h.set_option("symsweep", 1)
    elif variant == "full_sweep":  # (default = the library's choice: half sweep for exponential / spherical from 24 block columns on)
        h.set_option("symsweep", 0)
    elif variant == "pivoted":
        h.set_option("factor", 2)
    z, ss = m.execute("grid", *fx.grid_args(g), backend="vectorized")
    dz = float(np.abs(np.ma.getdata(z) - g["z"]).max())
    ds = float(np.abs(np.ma.getdata(ss) - g["ss"]).max())
    print("%s [%s]: N=%d, %d points, cond_1(A)=%.3g, max|dz|=%.3e, max|dss|=%.3e, invert %.2f ms" % (
        name, variant, g["x"].size, g["z"].size, float(g["cond1"]), dz, ds, m.last_timing["invert_ms"]))
    assert dz <= Z_TOL and ds <= SS_TOL
    nodes = _node_points(g)
    np.testing.assert_allclose(np.ma.getdata(z).ravel()[nodes], g["v"][:8], rtol=0, atol=Z_TOL)
    if "z_c" in g and variant == "default":
        zc, sc = m.execute("grid", *fx.grid_args(g), backend="C")
        assert type(zc) is np.ndarray
        assert np.abs(zc - g["z_c"]).max() <= Z_TOL and np.abs(sc - g["ss_c"]).max() <= SS_TOL


@pytest.mark.gpu
@pytest.mark.parametrize("window", [10, 100])
def test_hip_moving_window_matches_the_reference_at_config2_size(window):
    import pykrige_amd as pa

    g = _full("mw_c2")
    m = pa.OrdinaryKriging(g["x"], g["y"], g["v"], variogram_model=str(g["model"]), variogram_parameters=g["params_user"].tolist())
    for backend in ("C", "loop"):
        z, ss = m.execute("grid", g["gridx"], g["gridy"], backend=backend, n_closest_points=window)
        assert np.abs(np.asarray(z) - g["z_k%d" % window]).max() <= Z_TOL
        assert np.abs(np.asarray(ss) - g["ss_k%d" % window]).max() <= SS_TOL


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["asc", "descy", "descxy", "perm"])
def test_hip_external_z_on_any_axis_order(variant):
    import pykrige_amd as pa

    g = _r2()
    uk = pa.UniversalKriging(g["zs_x"], g["zs_y"], g["zs_v"], variogram_model="exponential",
                             variogram_parameters=[1.0, 0.3, 0.02], drift_terms=["external_Z"],
                             external_drift=g["zs_%s_dem" % variant], external_drift_x=g["zs_%s_ax" % variant],
                             external_drift_y=g["zs_%s_ay" % variant])
    z, ss = uk.execute("points", g["zs_qx"], g["zs_qy"], backend="vectorized")
    np.testing.assert_allclose(np.ma.getdata(z), g["zs_%s_z" % variant], rtol=0, atol=Z_TOL)
    np.testing.assert_allclose(np.ma.getdata(ss), g["zs_%s_ss" % variant], rtol=0, atol=SS_TOL)


@pytest.mark.gpu
def test_hip_update_variogram_model_sequences():
    import pykrige_amd as pa

    g = _r2()
    gx, gy, gz = g["upd_gx"], g["upd_gy"], g["upd_gz"]
    ok = pa.OrdinaryKriging(g["upd_x"], g["upd_y"], g["upd_v"], variogram_model="exponential",
                            variogram_parameters=[1.0, 0.3, 0.0], anisotropy_scaling=3.0, anisotropy_angle=45.0)
    ok.update_variogram_model("spherical", [1.0, 0.5, 0.05])
    z, ss = ok.execute("grid", gx, gy)
    np.testing.assert_allclose(np.ma.getdata(z), g["upd_ok_reset_z"], rtol=0, atol=Z_TOL)
    np.testing.assert_allclose(np.ma.getdata(ss), g["upd_ok_reset_ss"], rtol=0, atol=SS_TOL)
    ok.update_variogram_model("spherical", [1.0, 0.5, 0.05], anisotropy_scaling=2.0, anisotropy_angle=20.0)
    z, ss = ok.execute("grid", gx, gy)
    np.testing.assert_allclose(np.ma.getdata(z), g["upd_ok_set_z"], rtol=0, atol=Z_TOL)
    np.testing.assert_allclose(np.ma.getdata(ss), g["upd_ok_set_ss"], rtol=0, atol=SS_TOL)
    uk = pa.UniversalKriging(g["upd_x"], g["upd_y"], g["upd_v"], variogram_model="exponential",
                             variogram_parameters=[1.0, 0.3, 0.01], drift_terms=["regional_linear", "point_log"],
                             point_drift=g["upd_wells"], anisotropy_scaling=2.0, anisotropy_angle=30.0)
    uk.update_variogram_model("exponential", [1.0, 0.3, 0.01])
    z, ss = uk.execute("grid", gx, gy)
    np.testing.assert_allclose(np.ma.getdata(z), g["upd_uk_reset_z"], rtol=0, atol=Z_TOL)
    np.testing.assert_allclose(np.ma.getdata(ss), g["upd_uk_reset_ss"], rtol=0, atol=SS_TOL)
    k3 = pa.OrdinaryKriging3D(g["upd_x3"], g["upd_y3"], g["upd_z3"], g["upd_v3"], variogram_model="gaussian",
                              variogram_parameters=[1.0, 0.4, 0.02], anisotropy_scaling_y=1.5, anisotropy_scaling_z=2.0,
                              anisotropy_angle_x=10.0, anisotropy_angle_y=20.0, anisotropy_angle_z=30.0)
    k3.update_variogram_model("gaussian", [1.0, 0.4, 0.02], anisotropy_scaling_z=2.0)
    z, ss = k3.execute("grid", gx, gy, gz)
    np.testing.assert_allclose(np.ma.getdata(z), g["upd_ok3d_z"], rtol=0, atol=Z_TOL)
    np.testing.assert_allclose(np.ma.getdata(ss), g["upd_ok3d_ss"], rtol=0, atol=SS_TOL)


@pytest.mark.gpu
def test_hip_grid_style_with_a_mask_kriges_every_cell():
    g = fx.load("ok2d_masked_points")
    m = fx.amd_model_from("ok2d_masked_points", g)
    z0, s0 = m.execute("grid", g["gridx"], g["gridy"], backend="loop")
    z1, s1 = m.execute("grid", g["gridx"], g["gridy"], mask=g["mask"], backend="loop")
    z2, s2 = m.execute("grid", g["gridx"], g["gridy"], mask=g["mask"].T, backend="loop")
    assert np.array_equal(z0, z1) and np.array_equal(s0, s1) and np.array_equal(z0, z2)
    assert np.all(s1[g["mask"]] != 0.0)


@pytest.mark.gpu
@pytest.mark.parametrize("model", ["linear", "exponential", "spherical"])
def test_hip_statistics_with_pseudo_inverse(model):
    from pykrige_amd import core
    from pykrige_amd import variogram_models as vm

    g = _r2()
    fn = {"linear": vm.linear_variogram_model, "exponential": vm.exponential_variogram_model,
          "spherical": vm.spherical_variogram_model}[model]
    X = np.stack([g["pst_x"], g["pst_y"]], 1)
    d, s, e = core._find_statistics(X, g["pst_v"], fn, g["pst_%s_par" % model].tolist(), "euclidean", True)
    assert d.shape == g["pst_%s_delta" % model].shape
    np.testing.assert_allclose(d, g["pst_%s_delta" % model], rtol=0, atol=1e-7)
    np.testing.assert_allclose(s, g["pst_%s_sigma" % model], rtol=0, atol=1e-7)
    np.testing.assert_allclose(e, g["pst_%s_eps" % model], rtol=1e-6, atol=1e-6)


@pytest.mark.gpu
def test_hip_class_statistics_with_pseudo_inverse():
    import pykrige_amd as pa

    g = _r2()
    ok = pa.OrdinaryKriging(g["pst_x"], g["pst_y"], g["pst_v"], variogram_model="linear", variogram_parameters=[1.5, 0.0],
                            pseudo_inv=True, enable_statistics=True)
    np.testing.assert_allclose([ok.Q1, ok.Q2, ok.cR], [g["pst_class_Q1"], g["pst_class_Q2"], g["pst_class_cR"]], rtol=1e-6)


@pytest.mark.gpu
def test_hip_factor_is_reused_while_the_problem_is_unchanged():
    """ok.py:898 / 663 re-assemble and re-invert on every execute(); the drop-in keeps the factored matrix on the device."""
    g = fx.load("ok2d_n2000")
    m = fx.amd_model_from("ok2d_n2000", g)
    z1, s1 = m.execute("grid", g["gridx"], g["gridy"], backend="loop")
    assert m.factor_reused is False
    z2, s2 = m.execute("points", g["x"][:50] + 0.001, g["y"][:50], backend="loop")  # new points, same problem
    assert m.factor_reused is True and m.last_timing["invert_ms"] > 0.0
    z3, s3 = m.execute("grid", g["gridx"], g["gridy"], backend="loop")
    assert m.factor_reused is True and np.array_equal(z1, z3) and np.array_equal(s1, s3)
    m.update_variogram_model("exponential", [1.0, 0.35, 0.0])  # a different variogram: new factor
    z4, _ = m.execute("grid", g["gridx"], g["gridy"], backend="loop")
    assert m.factor_reused is False and np.abs(z4 - z1).max() > 1e-6
    m.update_variogram_model("exponential", [1.0, 0.3, 0.0])
    m._get_handle().set_option("factor", 2)  # a library option changed: no reuse
    z5, s5 = m.execute("grid", g["gridx"], g["gridy"], backend="loop")
    assert m.factor_reused is False and np.abs(z5 - g["z"]).max() <= Z_TOL and np.abs(s5 - g["ss"]).max() <= SS_TOL
    m.execute("grid", g["gridx"], g["gridy"], backend="loop", n_closest_points=8)  # the moving window replaces the handle's problem
    z6, s6 = m.execute("grid", g["gridx"], g["gridy"], backend="loop")
    assert m.factor_reused is False and np.abs(z6 - g["z"]).max() <= Z_TOL


@pytest.mark.gpu
def test_hip_masked_points_outside_the_external_drift_grid():
    """uk.py:967-971 looks the external drift up at EVERY point in the vectorized backend (so a masked point outside the drift
    grid raises there too) but only at the unmasked ones in the loop backend (uk.py:1034, 1061-1066)."""
    import pykrige_amd as pa

    g = _r2()
    uk = pa.UniversalKriging(g["zs_x"], g["zs_y"], g["zs_v"], variogram_model="exponential",
                             variogram_parameters=[1.0, 0.3, 0.02], drift_terms=["external_Z"], external_drift=g["zs_asc_dem"],
                             external_drift_x=g["zs_asc_ax"], external_drift_y=g["zs_asc_ay"])
    gx, gy = np.linspace(0.0, 1.4, 8), np.linspace(0.0, 1.0, 5)  # the last two columns lie outside the drift grid (x <= 1.1)
    mask = np.zeros((5, 8), dtype=bool)
    mask[:, 6:] = True
    with pytest.raises(ValueError, match="does not cover"):
        uk.execute("masked", gx, gy, mask=mask, backend="vectorized")
    z, ss = uk.execute("masked", gx, gy, mask=mask, backend="loop")
    zin, sin_ = uk.execute("grid", gx[:6], gy, backend="loop")
    assert np.array_equal(np.ma.getdata(z)[:, :6], zin) and np.array_equal(np.ma.getdata(ss)[:, :6], sin_)
    assert np.all(np.ma.getdata(z)[:, 6:] == 0.0) and z.mask[:, 6:].all()


# ------------------------------------------------------------------------------------------- GPU: the probe of the inverse
@pytest.mark.gpu
def test_ill_conditioned_inverse_is_caught_by_the_probe():
    """mik_factor probes every inverse it computes (include/mikrige.h, option "verify"): A c against the data vector, X A e_j
    against e_j.  A benign exponential model at N = 3200 keeps the half sweep the library chose (one attempt); the same
    stations with a range of 30 domain sizes (cond ~ 1e11) fail the probe twice -- half sweep, full sweep -- and end on
    partial pivoting, which is what LAPACK does for the reference (scipy.linalg.inv, ok.py:663)."""
    import pykrige_amd as pa

    rng = np.random.default_rng(100)
    n = 3200
    x, y = rng.random(n), rng.random(n)
    v = np.sin(6 * x) * np.cos(4 * y) + 0.1 * rng.standard_normal(n)
    gx, gy = np.linspace(0, 1, 23), np.linspace(0, 1, 19)
    for rng_par, want_attempts, want_path, want_half in ((0.3, 1, 1, 1), (30.0, 3, 2, 0)):
        st = ko.KrigingState(ndim=2, coords_orig=np.stack([x, y], 1), values=v, model="exponential",
                             params=ko.internal_parameters("exponential", [1.0, rng_par, 0.0]))
        zr, sr = ko.execute(st, "grid", gx, gy)
        a = ko.kriging_matrix(st)
        cond = float(np.linalg.cond(a, 1))
        ok = pa.OrdinaryKriging(x, y, v, variogram_model="exponential", variogram_parameters=[1.0, rng_par, 0.0])
        z, ss = ok.execute("grid", gx, gy, backend="loop")
        t = ok.last_timing
        dz, ds = float(np.abs(z - zr).max()), float(np.abs(ss - sr).max())
        print("range %.1f: cond_1 %.2e attempts %d path %d half %d res_z %.1e res_inv %.1e  |dz| %.1e |dss| %.1e verify %.2f ms"
              % (rng_par, cond, t["factor_attempts"], t["factor_path"], t["half_sweep"], t["verify_res_z"], t["verify_res_inv"], dz, ds, t["verify_ms"]))
        assert (t["factor_attempts"], t["factor_path"], t["half_sweep"]) == (want_attempts, want_path, want_half)
        assert dz <= max(Z_TOL, cond * 1e-15) and ds <= max(SS_TOL, cond * 1e-15)
        if want_attempts > 1:  # without the probe the sweep's answer would have been used: measurably worse
            ok._get_handle().set_option("verify", 0)
            ok._get_handle().set_option("symsweep", 0)
            z0, ss0 = ok.execute("grid", gx, gy, backend="loop")
            assert ok.last_timing["factor_attempts"] == 1 and ok.last_timing["factor_path"] == 1
            assert float(np.abs(ss0 - sr).max()) > 10 * ds
