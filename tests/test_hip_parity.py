"""GPU parity tests: the HIP path (through the C ABI, via the pykrige_amd classes or the raw handle)
against (a) golden vectors computed by the real reference (tests/golden, PyKrige 1.7.3
backend='vectorized') and (b) the CPU oracle on seeded synthetic inputs.

Tolerances are BASELINE.json's: |dz| <= 1e-8 and |dsigma^2| <= 1e-6 (absolute, on O(1) fields; for the
two reference fixtures whose values are O(100-1000) the z tolerance is scaled by max|z|)."""
import numpy as np
import pytest

from oracle import kriging_oracle as ko
from tests import _fixtures as fx

pytestmark = pytest.mark.gpu

Z_TOL, SS_TOL = 1e-8, 1e-6


def _lib():
    from pykrige_amd import _lib

    return _lib


def test_library_sees_a_gpu_and_mfma_layout():
    lib = _lib()
    assert lib.load().mik_device_count() >= 1
    lib.selftest_mfma(0)  # raises on a fragment-layout mismatch


def _handle_for(st, **opts):
    lib = _lib()
    h = lib.Handle(0)
    for k, v in opts.items():
        h.set_option(k, v)
    wells = st.wells_adj
    extra = []
    extra += list(st.specified_data)  # for "dem" fixtures the first entry is the external_Z drift at the stations
    extra += [f(*[st.coords_adj[:, k] for k in range(st.ndim)]) for f in st.functional]
    h.set_problem(ndim=st.ndim, xs=st.coords_adj[:, 0], ys=st.coords_adj[:, 1],
                  zs=st.coords_adj[:, 2] if st.ndim == 3 else None, values=st.values,
                  model_id=lib.MODEL_IDS[st.model], params=st.params, exact_values=st.exact_values,
                  regional_linear=st.regional_linear, wells=wells, extra_cols=np.array(extra) if extra else None,
                  geographic=st.geographic)
    return h


@pytest.mark.parametrize("name", [n for n in fx.names() if "A" in fx.load(n)])
def test_kriging_matrix_matches_reference(name):
    g = fx.load(name)
    st = fx.state_from(name, g)
    h = _handle_for(st)
    h.assemble_only()
    a = h.get_matrix(0)
    np.testing.assert_allclose(a, g["A"], rtol=0, atol=1e-12 * max(1.0, np.abs(g["A"]).max()))


@pytest.mark.parametrize("factor", [1, 2])
@pytest.mark.parametrize("name", ["ok2d_exponential_exact", "ok2d_spherical_exact", "uk2d_rl_pl", "ok3d_gaussian_aniso",
                                  "ok2d_n2000"])
def test_device_inverse_matches_lapack(name, factor):
    import scipy.linalg

    g = fx.load(name)
    st = fx.state_from(name, g)
    h = _handle_for(st, factor=factor)
    h.factor()
    assert h.timing()["factor_path"] == factor
    ainv = h.get_matrix(1)
    a = ko.kriging_matrix(st)
    ref = scipy.linalg.inv(a)
    # forward error of an inverse ~ cond * eps * |A^-1|
    scale = np.abs(ref).max()
    assert np.abs(ainv - ref).max() <= 1e-9 * scale
    resid = np.abs(a @ ainv - np.eye(a.shape[0])).max()
    assert resid <= 1e-8


@pytest.mark.parametrize("factor", [0, 1, 2])
@pytest.mark.parametrize("name", ["ok2d_linear_exact", "ok2d_power_exact", "ok2d_holeeffect_exact", "ref_test_ok3d"])
def test_factor_paths_agree_on_unbounded_and_hole_effect_models(name, factor):
    """linear / power have no sill: auto shifts by gamma(bounding-box diagonal) and sweeps; hole-effect may
    fail the positive-definiteness check and fall back to the pivoted path.  Every path must give the
    reference's numbers."""
    g = fx.load(name)
    m = fx.amd_model_from(name, g)
    m._get_handle().set_option("factor", factor)
    try:
        z, ss = m.execute("grid", *fx.grid_args(g), backend="loop")
    except np.linalg.LinAlgError:
        assert factor == 1  # a forced sweep may legitimately refuse a matrix that is not positive definite
        return
    assert m.last_timing["factor_path"] in ((1, 2) if factor == 0 else (factor,))
    zs = max(1.0, float(np.abs(g["z"]).max()))
    np.testing.assert_allclose(z, g["z"], rtol=0, atol=Z_TOL * zs)
    np.testing.assert_allclose(ss, g["ss"], rtol=0, atol=SS_TOL * max(1.0, float(np.abs(g["ss"]).max())))


@pytest.mark.parametrize("name", [n for n in fx.names() if "z" in fx.load(n)])
def test_execute_grid_matches_reference(name):
    g = fx.load(name)
    m = fx.amd_model_from(name, g)
    kw = {}
    if "spec_grid" in g:
        kw["specified_drift_arrays"] = [g["spec_grid"]]
    z, ss = m.execute("grid", *fx.grid_args(g), backend="vectorized", **kw)
    assert isinstance(z, np.ma.MaskedArray) and z.shape == g["z"].shape  # the reference's 'vectorized' return type
    zscale = max(1.0, float(np.abs(g["z"]).max()))
    sscale = max(1.0, float(np.abs(g["ss"]).max()))
    np.testing.assert_allclose(np.ma.getdata(z), g["z"], rtol=0, atol=Z_TOL * zscale)
    np.testing.assert_allclose(np.ma.getdata(ss), g["ss"], rtol=0, atol=SS_TOL * sscale)
    z2, ss2 = m.execute("grid", *fx.grid_args(g), backend="loop", **kw)
    assert type(z2) is np.ndarray
    np.testing.assert_array_equal(z2, np.ma.getdata(z))


def test_external_known_answers():
    """KT3D_H2O / KT3D grids the reference's own tests pin (tests/test_core.py:490-507, 707-725, 1914-1989)."""
    g = fx.load("ref_test_ok")
    z, _ = fx.amd_model_from("ref_test_ok", g).execute("grid", g["gridx"], g["gridy"], backend="loop")
    np.testing.assert_allclose(z, g["answer"], rtol=1e-5, atol=1e-8)
    g = fx.load("ref_test_uk")
    z, _ = fx.amd_model_from("ref_test_uk", g).execute("grid", g["gridx"], g["gridy"], backend="loop")
    np.testing.assert_allclose(z, g["answer"], rtol=1e-5, atol=1e-8)
    g = fx.load("ref_test_ok3d")
    z, ss = fx.amd_model_from("ref_test_ok3d", g).execute("grid", g["gridx"], g["gridy"], g["gridz"], backend="loop")
    np.testing.assert_allclose(z, g["answer_z"], rtol=1e-3, atol=1e-8)
    np.testing.assert_allclose(ss, g["answer_ss"], rtol=1e-3, atol=1e-8)


def test_external_drift_known_answer_and_pseudo_inverse():
    g = fx.load("ref_test_uk_external")  # tests/test_core.py:1479-1507
    z, _ = fx.amd_model_from("ref_test_uk_external", g).execute("grid", g["gridx"], g["gridy"], backend="loop")
    np.testing.assert_allclose(z, g["answer"], rtol=1e-5, atol=1e-8)
    import pykrige_amd as pa

    g = fx.load("pseudo_dup")  # duplicated stations (tests/test_core.py:2913-2949); reference outputs for both P_INV types
    d = g["d"]
    for p_type in ("pinv", "pinvh"):
        ok = pa.OrdinaryKriging(d[:, 0], d[:, 1], d[:, 2], variogram_parameters=[1.0, 0.0], pseudo_inv=True,
                                pseudo_inv_type=p_type)
        z, ss = ok.execute("grid", np.linspace(0, 1, 5), np.linspace(0, 1, 4), backend="loop")
        assert ok.last_timing["factor_path"] in (4, 5)  # pseudo-inverse on the device (deflated inverse or Jacobi), not a host SVD
        np.testing.assert_allclose(z, g["z_" + p_type], rtol=0, atol=1e-8)
        np.testing.assert_allclose(ss, g["ss_" + p_type], rtol=0, atol=1e-6)
        z1, _ = ok.execute("points", 0.0, 0.0, backend="loop")
        assert np.isclose(z1.item(), 2.0)  # mean of the redundant data


@pytest.mark.parametrize("fast", [1, 0])
@pytest.mark.parametrize("n,drift", [(301, False), (257, True)])
def test_device_pseudo_inverse_against_scipy(n, drift, fast):
    """mik_problem.pseudo_inv: the Moore-Penrose pseudo-inverse computed on the device (deflated regular inverse, or cyclic
    one-sided Jacobi with cut-off M eps sigma_max) against scipy.linalg.pinv of the same kriging matrix -- odd matrix orders (tournament padding),
    several duplicated stations (rank deficiency > 1), with and without drift rows -- and the kriging through it
    against the oracle fed with SciPy's pseudo-inverse (ok.py:660-661, uk.py:932-933)."""
    import scipy.linalg

    import pykrige_amd as pa

    (x, y), v = fx.synth(4000 + n, n, 2)
    x[-6:], y[-6:] = x[:6], y[:6]  # six duplicated stations with different values
    rng = np.random.default_rng(n)
    pts = rng.random((200, 2))
    user = [1.0, 0.5, 0.0]
    kw = dict(variogram_model="exponential", variogram_parameters=user, pseudo_inv=True, pseudo_inv_type="pinv")
    m = pa.UniversalKriging(x, y, v, drift_terms=["regional_linear"], **kw) if drift else pa.OrdinaryKriging(x, y, v, **kw)
    # fast = 1: the regular inverse of the matrix deflated by the duplicated stations' null space, verified with probe
    # vectors (factor_path 5); fast = 0: the general one-sided Jacobi pseudo-inverse (factor_path 4)
    m._get_handle().set_option("pinv_fast", fast)
    z, ss = m.execute("points", pts[:, 0], pts[:, 1], backend="loop")
    assert m.last_timing["factor_path"] == (5 if fast else 4)
    st = ko.KrigingState(ndim=2, coords_orig=np.stack([x, y], 1), values=v, model="exponential",
                         params=ko.internal_parameters("exponential", user), regional_linear=drift)
    a = ko.kriging_matrix(st)
    assert np.linalg.matrix_rank(a) <= a.shape[0] - 6
    pinv = scipy.linalg.pinv(a)
    got = m._get_handle().get_matrix(1)
    assert np.abs(got - pinv).max() <= 1e-9 * np.abs(pinv).max()
    zr, sr = ko.solve_points(st, ko.adjust_for_anisotropy(pts.copy(), st.center, st.scaling, st.angle), a_inv=pinv)
    np.testing.assert_allclose(z, zr, rtol=0, atol=Z_TOL)
    np.testing.assert_allclose(ss, sr, rtol=0, atol=SS_TOL)


@pytest.mark.parametrize("case", ["duplicates", "collinear_drift", "both"])
def test_block_jacobi_pseudo_inverse_against_scipy_and_the_scalar_form(case):
    """Round 4: the general pseudo-inverse as a BLOCK one-sided Jacobi (k_bj_gram / k_bj_eig / k_bj_rotate, option pinv_block = 1, the default
    from 1536 rows on) on the hard inputs of the CPU prototype -- duplicated stations with a zero nugget, collinear stations under a
    regional-linear drift, both (cond 5e7 on the range) -- against scipy.linalg.pinv (core.py:33 P_INV) and the scalar form; odd orders
    (block padding), ranks M - 1 .. M - 6."""
    import scipy.linalg

    lib = _lib()
    rng = np.random.default_rng(12)
    if case == "duplicates":
        (x, y), v = fx.synth(4301, 301, 2)
        x[-6:], y[-6:] = x[:6], y[:6]
        kw, okw, par = {}, {}, [1.0, 0.5, 0.0]
    else:
        n = 350
        x = rng.random(n)
        if case == "both":
            x[-3:] = x[:3]
        y, v = 0.5 * x - 0.1, np.cos(3 * x)
        kw, okw, par = dict(regional_linear=True), dict(regional_linear=True), ([1.0, 0.5, 0.0] if case == "both" else [1.0, 0.5, 0.02])
    st = ko.KrigingState(ndim=2, coords_orig=np.stack([x, y], 1), values=v, model="exponential", params=ko.internal_parameters("exponential", par), **okw)
    ref = scipy.linalg.pinv(ko.kriging_matrix(st))
    got = {}
    for block in (1, 0):
        h = lib.Handle(0)
        h.set_option("pinv_fast", 0)
        h.set_option("pinv_block", block)
        h.set_problem(ndim=2, xs=x, ys=y, zs=None, values=v, model_id=lib.MODEL_IDS["exponential"], params=st.params, pseudo_inv=1, **kw)
        h.factor()
        assert h.timing()["factor_path"] == 4
        got[block] = h.get_matrix(1)
        h.close()
    scale = np.abs(ref).max()
    assert np.abs(got[1] - ref).max() <= 1e-9 * scale, np.abs(got[1] - ref).max() / scale
    assert np.abs(got[1] - got[0]).max() <= 1e-9 * scale


def test_block_sweep_variants_agree():
    """Schedules of the unpivoted block sweep that are still in the library (round 6 removed the scalar-pivot diagonal kernels, the
    one-block / flag-ordered early-diagonal modes, the panel-row and tile-map variants with their options): the plain loop, the
    look-ahead (early-diagonal) schedule with and without the gate and the panel stream, odd steps walked backwards or not, return
    the bit-identical inverse; the half (upper-triangle) sweep an exactly symmetric one within 1e-11 of it."""
    g = fx.load("ok2d_n2000")
    st = fx.state_from("ok2d_n2000", g)
    h = _handle_for(st)
    h.set_option("factor", 1)
    h.set_option("symmetrize", 0)  # the full sweep as eliminated: schedules must agree bit for bit before the triangles are averaged
    ref = None
    # (look-ahead, half sweep, gate, panel stream, update_rev)
    for la, sym, gate, ps, rev in ((0, 0, 1, 0, 0), (1, 0, 1, 0, 0), (1, 0, 0, 0, 0), (1, 0, 1, 1, 0), (1, 0, 0, 1, 1), (0, 0, 1, 0, 1), (-1, 0, -1, -1, -1),
                                   (0, 1, 1, 0, 0), (1, 1, 1, 0, 0), (1, 1, 0, 0, 1), (1, 1, 1, 1, 0), (1, 1, 0, 1, 1), (-1, 1, -1, -1, -1)):
        for key, val in (("lookahead", la), ("symsweep", sym), ("gate", gate), ("panel_stream", ps), ("update_rev", rev)):
            h.set_option(key, val)
        h.factor()
        a = h.get_matrix(1)
        if ref is None:
            ref = a
        elif not sym:
            np.testing.assert_array_equal(a, ref)
        else:
            assert np.array_equal(a, a.T)
            assert np.abs(a - ref).max() <= 1e-11 * np.abs(ref).max()
    # as eliminated the full sweep's triangles differ by rounding; the default returns their average (what the symmetric contraction
    # reads one triangle of)
    assert not np.array_equal(ref, ref.T)
    for key, val in (("lookahead", -1), ("symsweep", 0), ("gate", -1), ("panel_stream", -1), ("update_rev", -1), ("symmetrize", 1)):
        h.set_option(key, val)
    h.factor()
    np.testing.assert_array_equal(h.get_matrix(1), 0.5 * (ref + ref.T))


def test_options_that_left_the_library_are_refused():
    """Round 6 prune: the experiments of rounds 2-5 are no longer options of the library (tools/mik_k_experiments.h, DESIGN_HISTORY.md):
    asking for one is an error, not a silent no-op."""
    lib = _lib()
    h = lib.Handle(0)
    for key in ("update_deep", "update_tpb", "pivot256", "update_pf", "update_token", "wide_reserve", "wide_colstream", "engine", "waves", "update_waves",
                "panel_rows", "update_map", "diag", "early_diag", "fuse_chain", "sparse_ktile", "sparse_epilogue"):
        with pytest.raises(ValueError, match="unknown option"):
            h.set_option(key, 1)
    h.close()


def test_lean_exp_of_the_moving_window_set_up_is_within_an_ulp_and_a_half():
    """Round 5: k_mw_chol's matrix set-up evaluates the variogram's exponential with exp_neg_lean (mik_dev.h: Cody-Waite reduction, degree-13
    polynomial, 19 instructions) instead of the library's exp.  Over the whole range an exponential / gaussian variogram can produce -- 0 down
    to underflow -- it stays within 1.5 ulp of NumPy's exp; exp(0) = 1 exactly; no NaN at the ends."""
    lib = _lib()
    rng = np.random.default_rng(3)
    x = -np.concatenate([[0.0, 1e-300, 1e-17, 0.5 * np.log(2.0), np.log(2.0), 1.0, 708.0, 745.0, 745.2, 800.0, 1e6, 3e9, 1e300, np.inf],
                         10.0 ** rng.uniform(-12, 2.9, 200000), rng.uniform(0.0, 50.0, 200000)])
    got = lib.selftest_exp(x)
    ref = np.exp(x)
    assert got[0] == 1.0 and np.all(np.isfinite(got)) and np.all(got >= 0.0)
    normal = ref > 1e-300
    ulp = np.spacing(ref[normal])
    err = np.abs(got[normal] - ref[normal]) / ulp
    assert err.max() <= 1.5, err.max()
    assert np.abs(got[~normal] - ref[~normal]).max() <= 1e-300  # towards underflow: absolute
    # and where it is used: the LDL^T kernel with the model compiled in against the dynamic form (library exp, the reference's divisions)
    import pykrige_amd as pa

    g = fx.load("mw_ok2d")
    ext = float(max(np.ptp(g["x"]), np.ptp(g["y"])))
    for model, par in (("exponential", [1.0, 0.4 * ext, 0.02]), ("gaussian", [1.0, 0.3 * ext, 0.05]), ("spherical", [1.0, 0.5 * ext, 0.0])):
        outs = []
        for static in (1, 0):
            m = pa.OrdinaryKriging(g["x"], g["y"], g["v"], variogram_model=model, variogram_parameters=par)
            m._get_handle().set_option("mw_static", static)
            z, ss = m.execute("grid", g["gridx"], g["gridy"], backend="loop", n_closest_points=20)
            outs.append((np.ma.getdata(z).copy(), np.ma.getdata(ss).copy()))
        assert np.abs(outs[0][0] - outs[1][0]).max() <= 1e-11 and np.abs(outs[0][1] - outs[1][1]).max() <= 1e-11, model


def test_triangular_diagonal_blocks_of_the_contraction():
    """Round 3: the symmetric contraction takes the diagonal block of a tile as a triangle of 16-row groups (option tri): sigma^2
    equal to the whole-block form to rounding, z untouched (it is a separate dot product), for a station count that leaves a short last
    block and for one that does not."""
    for name in ("ok2d_n2000", "ok2d_exponential_exact"):
        g = fx.load(name)
        st = fx.state_from(name, g)
        out = []
        for tri in (0, 1):
            h = _handle_for(st)
            h.set_option("tri", tri)
            h.factor()
            rng = np.random.default_rng(5)
            n = 1000
            xs, ys = st.coords_adj[:, 0], st.coords_adj[:, 1]
            px = rng.uniform(xs.min(), xs.max(), n)
            py = rng.uniform(ys.min(), ys.max(), n)
            px[:8], py[:8] = xs[:8], ys[:8]  # exact hits: sigma^2 = 0 is a difference of large terms
            h.set_points(px, py)
            h.predict()
            out.append(tuple(np.array(v) for v in h.get_results()))
            h.close()
        np.testing.assert_array_equal(out[0][0], out[1][0])
        scale = max(1.0, float(np.abs(out[0][1]).max()))
        assert np.abs(out[0][1] - out[1][1]).max() <= 1e-11 * scale


def test_rccl_single_rank_broadcast_path():
    """The multi-GPU exchange with world size 1: dlopen(librccl), ncclCommInitRank, ncclBroadcast of the
    inverse + c on the handle's stream.  (More ranks cannot be had on a 1-GPU box; tests/test_dist_gloo.py
    covers the host protocol with world size 2.)"""
    lib = _lib()
    g = fx.load("ok2d_n2000")
    st = fx.state_from("ok2d_n2000", g)
    h = _handle_for(st)
    h.comm_init(1, 0, lib.Handle.comm_unique_id())
    h.factor()
    ref = h.get_matrix(1)
    h.bcast_factor(0)
    np.testing.assert_array_equal(h.get_matrix(1), ref)
    GX, GY = np.meshgrid(g["gridx"], g["gridy"])
    pts = ko.adjust_for_anisotropy(np.stack([GX.ravel(), GY.ravel()], 1), st.center, st.scaling, st.angle)
    h.set_points(pts[:, 0], pts[:, 1])
    h.predict()
    z, ss = h.get_results()
    np.testing.assert_allclose(z, g["z"].ravel(), rtol=0, atol=Z_TOL)
    np.testing.assert_allclose(ss, g["ss"].ravel(), rtol=0, atol=SS_TOL)


def test_masked_and_points_styles():
    g = fx.load("ok2d_masked_points")
    m = fx.amd_model_from("ok2d_masked_points", g)
    z, ss = m.execute("masked", g["gridx"], g["gridy"], mask=g["mask"], backend="loop")
    assert isinstance(z, np.ma.MaskedArray) and z.shape == g["mask"].shape
    keep = ~g["mask"]
    np.testing.assert_allclose(np.ma.getdata(z)[keep], g["z_masked"][keep], rtol=0, atol=Z_TOL)
    np.testing.assert_allclose(np.ma.getdata(ss)[keep], g["ss_masked"][keep], rtol=0, atol=SS_TOL)
    assert np.all(np.ma.getdata(z)[g["mask"]] == 0.0) and z[g["mask"]].mask.all()
    zt, _ = m.execute("masked", g["gridx"], g["gridy"], mask=g["mask"].T, backend="loop")  # auto-transpose
    np.testing.assert_array_equal(np.ma.getdata(zt), np.ma.getdata(z))
    zp, ssp = m.execute("points", g["px"], g["py"], backend="C")
    np.testing.assert_allclose(zp, g["z_points"], rtol=0, atol=Z_TOL)
    np.testing.assert_allclose(ssp, g["ss_points"], rtol=0, atol=SS_TOL)
    # first four points sit on stations: exact interpolation
    np.testing.assert_allclose(zp[:4], g["v"][:4], rtol=0, atol=Z_TOL)
    assert np.all(np.abs(ssp[:4]) <= SS_TOL)
    g3 = fx.load("ok3d_gaussian_aniso")
    m3 = fx.amd_model_from("ok3d_gaussian_aniso", g3)
    z3, ss3 = m3.execute("masked", g3["gridx"], g3["gridy"], g3["gridz"], mask=g3["mask"], backend="loop")
    keep = ~g3["mask"]
    np.testing.assert_allclose(np.ma.getdata(z3)[keep], g3["z_masked"][keep], rtol=0, atol=Z_TOL)
    np.testing.assert_allclose(np.ma.getdata(ss3)[keep], g3["ss_masked"][keep], rtol=0, atol=SS_TOL)


def test_errors_mirror_the_reference():
    import pykrige_amd as pa

    (x, y), v = fx.synth(7, 30, 2)
    ok = pa.OrdinaryKriging(x, y, v, variogram_model="exponential", variogram_parameters=[1.0, 0.3, 0.0])
    with pytest.raises(ValueError):
        ok.execute("blurg", [0.0, 1.0], [0.0, 1.0])
    with pytest.raises(IOError):
        ok.execute("masked", [0.0, 1.0], [0.0, 1.0])
    with pytest.raises(ValueError):
        ok.execute("masked", [0.0, 1.0], [0.0, 1.0, 2.0], mask=np.zeros((5, 5), bool))
    with pytest.raises(ValueError):
        ok.execute("points", [0.0, 1.0], [0.0, 1.0, 2.0])
    with pytest.raises(ValueError):
        ok.execute("grid", [0.0, 1.0], [0.0, 1.0], backend="mystery")
    z, ss = ok.execute("grid", np.arange(3), np.arange(2), backend="loop")  # integer axes are cast, not a crash
    assert z.shape == (2, 3)
    dup = pa.OrdinaryKriging([0.0, 0.0, 1.0], [0.0, 0.0, 1.0], [1.0, 1.0, 2.0], variogram_model="linear",
                             variogram_parameters=[1.0, 0.0])
    with pytest.raises(np.linalg.LinAlgError):  # duplicate stations: scipy.linalg.inv raises LinAlgError
        dup.execute("grid", [0.5], [0.5], backend="loop")
    pin = pa.OrdinaryKriging([0.0, 0.0, 1.0, 2.0], [0.0, 0.0, 1.0, 0.5], [1.0, 1.0, 2.0, 3.0], variogram_model="linear",
                             variogram_parameters=[1.0, 0.0], pseudo_inv=True)
    z, ss = pin.execute("grid", [0.5], [0.5], backend="loop")  # test_core.py:2913-2929: pinv averages duplicates
    assert np.isfinite(z).all()


@pytest.mark.parametrize("cfg", ["ok2d_exp", "ok2d_sph", "ok3d_gau", "uk2d_rl_pl"])
def test_synthetic_n1500_against_oracle(cfg):
    """Seeded SURVEY 8(d)-style inputs at a size the oracle finishes in seconds, with grid nodes on
    stations (eps rule), several contraction chunks (chunk=1024) and both symmetric settings."""
    import pykrige_amd as pa

    n = 1500
    if cfg == "ok3d_gau":
        (x, y, zc), v = fx.synth(3, n, 3)
        axes = [np.linspace(0, 1, 21), np.linspace(0, 1, 17), np.linspace(0, 1, 9)]
        for k in range(8):
            x[k], y[k], zc[k] = axes[0][2 * k], axes[1][k], axes[2][k]
        m = pa.OrdinaryKriging3D(x, y, zc, v, variogram_model="gaussian", variogram_parameters=[1.0, 0.4, 0.02])
        st = ko.KrigingState(ndim=3, coords_orig=np.stack([x, y, zc], 1), values=v, model="gaussian",
                             params=ko.internal_parameters("gaussian", [1.0, 0.4, 0.02]), scaling=[1.0, 1.0],
                             angle=[0.0, 0.0, 0.0])
    else:
        (x, y), v = fx.synth(2, n, 2)
        axes = [np.linspace(0, 1, 61), np.linspace(0, 1, 53)]
        for k in range(8):
            x[k], y[k] = axes[0][5 * k + 1], axes[1][3 * k + 2]
        if cfg == "uk2d_rl_pl":
            wells = [[0.3137, 0.7219, 1.0], [0.6621, 0.2483, -0.5], [0.8412, 0.8127, 2.0]]
            m = pa.UniversalKriging(x, y, v, variogram_model="exponential", variogram_parameters=[1.0, 0.3, 0.01],
                                    drift_terms=["regional_linear", "point_log"], point_drift=wells)
            st = ko.KrigingState(ndim=2, coords_orig=np.stack([x, y], 1), values=v, model="exponential",
                                 params=ko.internal_parameters("exponential", [1.0, 0.3, 0.01]), regional_linear=True,
                                 point_log=np.array(wells))
        else:
            model, par = ("exponential", [1.0, 0.3, 0.0]) if cfg == "ok2d_exp" else ("spherical", [1.0, 0.2, 0.01])
            m = pa.OrdinaryKriging(x, y, v, variogram_model=model, variogram_parameters=par)
            st = ko.KrigingState(ndim=2, coords_orig=np.stack([x, y], 1), values=v, model=model,
                                 params=ko.internal_parameters(model, par))
    zr, sr = ko.execute(st, "grid", *axes)
    h = m._get_handle()
    outs = []
    # the symmetric half product with triangular / whole diagonal blocks, and the reference's full product w = A_inv b (three kernels)
    for sym, tri in ((1, 1), (0, 1), (1, 0)):
        h.set_option("symmetric", sym)
        h.set_option("tri", tri)
        h.set_option("chunk", 1024)
        z, ss = m.execute("grid", *axes, backend="loop")
        assert m.last_timing["contract_launches"] >= 3 and m.last_timing["symmetric"] == sym
        np.testing.assert_allclose(z, zr, rtol=0, atol=Z_TOL)
        np.testing.assert_allclose(ss, sr, rtol=0, atol=SS_TOL)
        outs.append((z, ss))
    np.testing.assert_allclose(outs[0][1], outs[1][1], rtol=0, atol=1e-9)
    # exact-hit nodes: z == datum, sigma^2 == 0 (tests/test_core.py:1510-1834 semantics)
    pts = m.execute("points", x[:8], y[:8], *([zc[:8]] if cfg == "ok3d_gau" else []), backend="loop")
    np.testing.assert_allclose(pts[0], v[:8], rtol=0, atol=Z_TOL)
    assert np.all(np.abs(pts[1]) <= SS_TOL)


def test_full_size_config2_properties():
    """BASELINE config 2 at full size (N=5000, 1000x1000 grid, exponential [1,0.3,0]): size-independent
    properties + one 4096-point slab against the oracle."""
    import pykrige_amd as pa

    n = 5000
    (x, y), v = fx.synth(2, n, 2)
    gx = gy = np.linspace(0.0, 1.0, 1000)
    for k in range(8):
        x[k], y[k] = gx[100 * k + 7], gy[90 * k + 11]
    ok = pa.OrdinaryKriging(x, y, v, variogram_model="exponential", variogram_parameters=[1.0, 0.3, 0.0])
    z, ss = ok.execute("grid", gx, gy, backend="loop")
    assert z.shape == (1000, 1000) and np.isfinite(z).all() and np.isfinite(ss).all()
    # (1) exact interpolation at the 8 nodes that coincide with stations
    for k in range(8):
        assert abs(z[90 * k + 11, 100 * k + 7] - v[k]) <= Z_TOL
        assert abs(ss[90 * k + 11, 100 * k + 7]) <= SS_TOL
    # (2) kriging variance bounds: 0 <= sigma^2 <= sill + |mu| slack ; here simply within [ -tol, 2*sill ]
    assert ss.min() >= -SS_TOL and ss.max() <= 2.0
    # (3) linearity in the data: z(v + c) = z(v) + c for a constant shift (weights sum to one)
    ok2 = pa.OrdinaryKriging(x, y, v + 3.25, variogram_model="exponential", variogram_parameters=[1.0, 0.3, 0.0])
    zs, sss = ok2.execute("grid", gx, gy[500:504], backend="loop")
    np.testing.assert_allclose(zs, z[500:504] + 3.25, rtol=0, atol=Z_TOL)
    np.testing.assert_allclose(sss, ss[500:504], rtol=0, atol=1e-9)
    # (4) a 4-row slab against the CPU oracle
    st = ko.KrigingState(ndim=2, coords_orig=np.stack([x, y], 1), values=v, model="exponential",
                         params=ko.internal_parameters("exponential", [1.0, 0.3, 0.0]))
    zr, sr = ko.execute(st, "grid", gx, gy[500:504])
    np.testing.assert_allclose(z[500:504], zr, rtol=0, atol=Z_TOL)
    np.testing.assert_allclose(ss[500:504], sr, rtol=0, atol=SS_TOL)


@pytest.mark.parametrize("name", ["mw_ok2d", "mw_ok3d", "geo_ok2d"])
def test_moving_window_matches_reference(name):
    """n_closest_points (ok.py:929-986, cok.pyx:98-193): kNN + per-point (k+1)x(k+1) solve on the device vs the
    real reference's backend='loop' outputs, for k = 2 .. 70 (16/32/64/256 threads per point)."""
    g = fx.load(name)
    m = fx.amd_model_from(name, g)
    axes = fx.grid_args(g)
    for key in [k for k in g if k.startswith("z_k")]:
        k = int(key[3:])
        for backend in (("loop", "C") if name != "mw_ok3d" else ("loop",)):
            z, ss = m.execute("grid", *axes, backend=backend, n_closest_points=k)
            assert type(z) is np.ndarray and z.shape == g[key].shape
            np.testing.assert_allclose(z, g[key], rtol=0, atol=Z_TOL)
            np.testing.assert_allclose(ss, g["ss_k%d" % k], rtol=0, atol=SS_TOL)
    if name == "mw_ok2d":
        # the HBM-resident neighbour lists (taken on their own only when k + 256 > 8192) forced for small windows as well
        from pykrige_amd import _lib as lib

        old = lib.Handle.set_problem

        def forced(self, *a, **kw):
            self.set_option("mw_lds_cap", 0)
            return old(self, *a, **kw)

        lib.Handle.set_problem = forced
        try:
            m2 = fx.amd_model_from(name, g)
            for k in (10, 200):
                z, ss = m2.execute("grid", *axes, backend="loop", n_closest_points=k)
                np.testing.assert_allclose(z, g["z_k%d" % k], rtol=0, atol=Z_TOL)
                np.testing.assert_allclose(ss, g["ss_k%d" % k], rtol=0, atol=SS_TOL)
        finally:
            lib.Handle.set_problem = old
        # the pivoted variant of the per-point solve (default: SPD-shifted without pivot search) on the same fixtures
        m._get_handle().set_option("mw_pivot", 1)
        for k in (2, 10, 31, 70):
            z, ss = m.execute("grid", *axes, backend="loop", n_closest_points=k)
            np.testing.assert_allclose(z, g["z_k%d" % k], rtol=0, atol=Z_TOL)
            np.testing.assert_allclose(ss, g["ss_k%d" % k], rtol=0, atol=SS_TOL)
        m._get_handle().set_option("mw_pivot", 0)
        zm, ssm = m.execute("masked", *axes, mask=g["mask"], backend="loop", n_closest_points=10)
        keep = ~g["mask"]
        np.testing.assert_allclose(np.ma.getdata(zm)[keep], g["zm_k10"][keep], rtol=0, atol=Z_TOL)
        np.testing.assert_allclose(np.ma.getdata(ssm)[keep], g["ssm_k10"][keep], rtol=0, atol=SS_TOL)
        # five stations sit on grid nodes: exact interpolation through the window as well
        z, ss = m.execute("points", g["x"][:5], g["y"][:5], backend="C", n_closest_points=10)
        np.testing.assert_allclose(z, g["v"][:5], rtol=0, atol=Z_TOL)
        assert np.all(np.abs(ss) <= SS_TOL)
        with pytest.raises(ValueError):  # ok.py:982-986
            m.execute("grid", *axes, backend="vectorized", n_closest_points=10)
        with pytest.raises(ValueError):
            m.execute("grid", *axes, backend="loop", n_closest_points=1)
        # after a moving-window call the ordinary path must still work (the matrix buffer was reused)
        z0, _ = m.execute("grid", *axes, backend="loop")
        zr, _ = ko.execute(fx.state_from(name, g), "grid", *axes)
        np.testing.assert_allclose(z0, zr, rtol=0, atol=Z_TOL)
    elif name == "mw_ok3d":
        with pytest.raises(ValueError):  # ok3d.py:906-912: only 'loop' takes a moving window
            m.execute("grid", *axes, backend="vectorized", n_closest_points=8)


def test_variogram_fit_statistics_match_reference():
    """core._find_statistics (N-1 growing solves on the CPU) vs the O(N^3) bordered-inverse recursion on the device:
    delta, sigma, epsilon and Q1/Q2/cR computed by the real reference (tests/golden/stats_find_statistics.npz)."""
    import pykrige_amd as pa

    g = fx.load("stats_find_statistics")
    cases = (("exp", "exponential", [1.0, 0.3, 0.1]), ("lin", "linear", [1.3, 0.05]), ("sph", "spherical", [0.85, 0.5, 0.05]))
    for tag, model, user in cases:  # user list form = full sill for the bounded models
        ok = pa.OrdinaryKriging(g["x"], g["y"], g["v"], variogram_model=model, variogram_parameters=user, enable_statistics=True)
        np.testing.assert_allclose(ok.delta, g["delta_" + tag], rtol=0, atol=1e-8)
        np.testing.assert_allclose(ok.sigma, g["sigma_" + tag], rtol=0, atol=1e-8)
        np.testing.assert_allclose(ok.get_epsilon_residuals(), g["eps_" + tag], rtol=0, atol=1e-7)
        np.testing.assert_allclose(ok.get_statistics(), g["q_" + tag], rtol=1e-7)
    # the 3-D / universal classes expose the same attributes (computed on first use, with the ordinary system)
    k3 = pa.UniversalKriging3D(g["x3"], g["y3"], g["z3"], g["v3"], variogram_model="gaussian", variogram_parameters=[1.0, 0.5, 0.1],
                               drift_terms=["regional_linear"])
    np.testing.assert_allclose(k3.sigma, g["sigma_3d"], rtol=0, atol=1e-8)
    np.testing.assert_allclose([k3.Q1, k3.Q2, k3.cR], g["q_3d"], rtol=1e-7)
    geo = pa.OrdinaryKriging(g["lon"], g["lat"], g["vg"], variogram_model="exponential", variogram_parameters=[1.0, 40.0, 0.1],
                             coordinates_type="geographic", enable_statistics=True)
    np.testing.assert_allclose(geo.delta, g["delta_geo"], rtol=0, atol=1e-8)
    np.testing.assert_allclose(geo.sigma, g["sigma_geo"], rtol=0, atol=1e-8)
    plain = pa.OrdinaryKriging(g["x"], g["y"], g["v"], variogram_model="linear", variogram_parameters=[1.3, 0.05])
    assert plain.Q1 is None and plain.delta is None  # ok.py:376-377: not computed unless enable_statistics
    dup = pa.OrdinaryKriging([0.0, 0.0, 1.0, 2.0], [0.0, 0.0, 1.0, 0.5], [1.0, 1.0, 2.0, 3.0], variogram_model="linear",
                             variogram_parameters=[1.0, 0.0])
    with pytest.raises(np.linalg.LinAlgError):  # coincident stations: np.linalg.solve in core._krige raises
        dup._compute_statistics()


def test_one_shot_c_entry_point_and_degenerate_inputs():
    """mik_krige_execute (the _c_exec_loop-shaped call) straight through ctypes; an all-masked grid; a single station;
    ragged sizes (1 point, 129 points = one full tile + 1)."""
    import ctypes as C

    import pykrige_amd as pa

    lib = _lib()
    L = lib.load()
    (x, y), v = fx.synth(21, 64, 2)
    rng = np.random.default_rng(22)
    px, py = rng.random(129), rng.random(129)
    mask = np.zeros(129, dtype=np.int8)
    mask[[0, 5, 128]] = 1
    p, g = lib.MikProblem(), lib.MikPoints()
    arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (x, y, v, px, py)]
    dp = C.POINTER(C.c_double)
    p.ndim, p.model_id, p.n = 2, lib.MODEL_IDS["exponential"], 64
    p.xs, p.ys, p.values = (a.ctypes.data_as(dp) for a in arrs[:3])
    p.params = (C.c_double * 3)(0.9, 0.3, 0.1)
    p.eps, p.exact_values = 1e-10, 1
    g.npt, g.px, g.py = 129, arrs[3].ctypes.data_as(dp), arrs[4].ctypes.data_as(dp)
    g.mask = mask.ctypes.data_as(C.POINTER(C.c_int8))
    z, ss = np.full(129, 7.0), np.full(129, 7.0)
    assert L.mik_krige_execute(0, C.byref(p), C.byref(g), z.ctypes.data_as(dp), ss.ctypes.data_as(dp)) == 0
    st = ko.KrigingState(ndim=2, coords_orig=np.stack([x, y], 1), values=v, model="exponential", params=[0.9, 0.3, 0.1])
    keep = mask == 0
    zr, sr = ko.solve_points(st, np.stack([px, py], 1)[keep])
    np.testing.assert_allclose(z[keep], zr, rtol=0, atol=Z_TOL)
    np.testing.assert_allclose(ss[keep], sr, rtol=0, atol=SS_TOL)
    assert np.all(z[~keep] == 0.0) and np.all(ss[~keep] == 0.0)  # masked points: outputs 0.0 (cok.pyx:25-26, 57-58)
    p.model_id = 99
    assert L.mik_krige_execute(0, C.byref(p), C.byref(g), z.ctypes.data_as(dp), ss.ctypes.data_as(dp)) == lib.MIK_EINVAL
    assert b"variogram" in L.mik_last_error()
    ok = pa.OrdinaryKriging(x, y, v, variogram_model="exponential", variogram_parameters=[1.0, 0.3, 0.1])
    zm, sm = ok.execute("masked", [0.1, 0.5], [0.2, 0.4, 0.9], mask=np.ones((3, 2), bool), backend="loop")
    assert zm.mask.all() and np.all(np.ma.getdata(zm) == 0.0) and zm.shape == (3, 2)
    z1, s1 = ok.execute("points", 0.3, 0.7, backend="loop")  # scalars, like krige.execute("points", 0.0, 0.0)
    zr1, sr1 = ko.solve_points(ko.KrigingState(ndim=2, coords_orig=np.stack([x, y], 1), values=v, model="exponential",
                                               params=[0.9, 0.3, 0.1]), np.array([[0.3, 0.7]]))
    assert z1.shape == (1,) and abs(z1[0] - zr1[0]) <= Z_TOL and abs(s1[0] - sr1[0]) <= SS_TOL
    one = pa.OrdinaryKriging([0.5], [0.5], [3.0], variogram_model="linear", variogram_parameters=[1.0, 0.2])
    zo, so = one.execute("points", [0.0, 0.5], [0.0, 0.5], backend="loop")  # one station: weight 1 everywhere
    np.testing.assert_allclose(zo, [3.0, 3.0], atol=1e-12)
    zro, sro = ko.solve_points(ko.KrigingState(ndim=2, coords_orig=np.array([[0.5, 0.5]]), values=np.array([3.0]), model="linear",
                                               params=[1.0, 0.2]), np.array([[0.0, 0.0], [0.5, 0.5]]))
    np.testing.assert_allclose(so, sro, atol=1e-12)


@pytest.mark.parametrize("cfg", [3, 4, 5])
def test_baseline_configs_at_full_station_count(cfg):
    """BASELINE configs 3-5 with their full station counts (N = 2000 / 4000 / 8000, SURVEY 8(d) seeds and variogram
    parameters) on a 1536-point sample of their grids, against the oracle; 8 sample points coincide with stations."""
    import pykrige_amd as pa

    rng = np.random.default_rng(100 + cfg)
    if cfg == 3:
        (x, y, zc), v = fx.synth(3, 2000, 3)
        m = pa.OrdinaryKriging3D(x, y, zc, v, variogram_model="gaussian", variogram_parameters=[1.0, 0.4, 0.02])
        st = ko.KrigingState(ndim=3, coords_orig=np.stack([x, y, zc], 1), values=v, model="gaussian",
                             params=ko.internal_parameters("gaussian", [1.0, 0.4, 0.02]), scaling=[1.0, 1.0], angle=[0.0] * 3)
        pts = rng.random((1536, 3))
        pts[:8] = np.stack([x[:8], y[:8], zc[:8]], 1)
        z, ss = m.execute("points", pts[:, 0], pts[:, 1], pts[:, 2], backend="loop")
    else:
        n = 4000 if cfg == 4 else 8000
        (x, y), v = fx.synth(cfg, n, 2)
        pts = rng.random((1536, 2))
        pts[:8] = np.stack([x[:8], y[:8]], 1)
        if cfg == 4:
            wells = [[0.3137, 0.7219, 1.0], [0.6621, 0.2483, -0.5], [0.8412, 0.8127, 2.0]]
            m = pa.UniversalKriging(x, y, v, variogram_model="exponential", variogram_parameters=[1.0, 0.3, 0.01],
                                    drift_terms=["regional_linear", "point_log"], point_drift=wells)
            st = ko.KrigingState(ndim=2, coords_orig=np.stack([x, y], 1), values=v, model="exponential",
                                 params=ko.internal_parameters("exponential", [1.0, 0.3, 0.01]), regional_linear=True,
                                 point_log=np.array(wells))
        else:
            m = pa.OrdinaryKriging(x, y, v, variogram_model="spherical", variogram_parameters=[1.0, 0.2, 0.01])
            st = ko.KrigingState(ndim=2, coords_orig=np.stack([x, y], 1), values=v, model="spherical",
                                 params=ko.internal_parameters("spherical", [1.0, 0.2, 0.01]))
        z, ss = m.execute("points", pts[:, 0], pts[:, 1], backend="loop")
    zr, sr = ko.solve_points(st, pts)
    np.testing.assert_allclose(z, zr, rtol=0, atol=Z_TOL)
    np.testing.assert_allclose(ss, sr, rtol=0, atol=SS_TOL)
    np.testing.assert_allclose(z[:8], v[:8], rtol=0, atol=Z_TOL)
    assert np.all(np.abs(ss[:8]) <= SS_TOL) and m.last_timing["factor_path"] == 1


def test_device_experimental_variogram_matches_reference_binning():
    """mik_experimental_variogram (pair distances + lag bins on the GPU) vs lags / semivariances the real reference
    computed (fit_variograms.npz, geo_ok2d.npz), and the fit that follows from them."""
    import pykrige_amd as pa
    from pykrige_amd import core

    lib = _lib()
    g = fx.load("fit_variograms")
    xy = core.adjust_for_anisotropy(np.stack([g["x"], g["y"]], 1), [(g["x"].max() + g["x"].min()) / 2, (g["y"].max() + g["y"].min()) / 2],
                                    [2.0], [30.0])
    h = lib.Handle(0)
    h.set_problem(ndim=2, xs=xy[:, 0], ys=xy[:, 1], zs=None, values=g["v"], model_id=0, params=[1.0, 0.0])
    lags, semi = h.experimental_variogram(8)
    np.testing.assert_allclose(lags, g["lags_linear_0"], rtol=1e-12)
    np.testing.assert_allclose(semi, g["semi_linear_0"], rtol=1e-12)
    gg = fx.load("geo_ok2d")
    h.set_problem(ndim=2, xs=gg["x"], ys=gg["y"], zs=None, values=gg["v"], model_id=0, params=[1.0, 0.0], geographic=True)
    lags, semi = h.experimental_variogram(7)
    np.testing.assert_allclose(lags, gg["fit_lags"], rtol=1e-11)
    np.testing.assert_allclose(semi, gg["fit_semi"], rtol=1e-11)
    import os

    os.environ["MIK_DEVICE_VARIOGRAM_MIN_N"] = "100"  # route the constructor's binning through the device
    try:
        ok = pa.OrdinaryKriging(g["x"], g["y"], g["v"], variogram_model="spherical", nlags=8, anisotropy_scaling=2.0,
                                anisotropy_angle=30.0)
        np.testing.assert_allclose(ok.variogram_model_parameters, g["par_spherical_0"], rtol=1e-6, atol=1e-9)
        k3 = pa.OrdinaryKriging3D(g["x3"], g["y3"], g["z3"], g["v3"], variogram_model="spherical", nlags=6)
        np.testing.assert_allclose(k3.variogram_model_parameters, g["par_3d"], rtol=1e-6, atol=1e-9)
        np.testing.assert_allclose(k3.lags, g["lags_3d"], rtol=1e-12)
    finally:
        del os.environ["MIK_DEVICE_VARIOGRAM_MIN_N"]


@pytest.mark.parametrize("n", [2, 3, 14, 15, 16, 17, 110, 126, 127, 128, 129, 254, 255, 256, 257, 383, 384, 385])
def test_matrix_orders_around_tile_boundaries(n):
    """M = n + 1 (OK) and n + 4 (UK with regional_linear + one well) straddle the 16-wide K tile and the 128-wide
    block boundaries of the device kernels; every factor path and both contraction forms against the oracle."""
    import pykrige_amd as pa

    (x, y), v = fx.synth(1000 + n, n, 2)
    rng = np.random.default_rng(n)
    pts = rng.random((131, 2))
    pts[0] = [x[0], y[0]]
    st = ko.KrigingState(ndim=2, coords_orig=np.stack([x, y], 1), values=v, model="spherical",
                         params=ko.internal_parameters("spherical", [1.0, 0.6, 0.05]))
    zr, sr = ko.solve_points(st, ko.adjust_for_anisotropy(pts, st.center, st.scaling, st.angle))
    ok = pa.OrdinaryKriging(x, y, v, variogram_model="spherical", variogram_parameters=[1.0, 0.6, 0.05])
    h = ok._get_handle()
    for factor, sym in ((0, 1), (2, 0), (1, 0), (2, 1)):
        h.set_option("factor", factor)
        h.set_option("symmetric", sym)
        z, ss = ok.execute("points", pts[:, 0], pts[:, 1], backend="loop")
        np.testing.assert_allclose(z, zr, rtol=0, atol=Z_TOL)
        np.testing.assert_allclose(ss, sr, rtol=0, atol=SS_TOL)
    if n >= 14:
        well = [[0.37, 0.61, 1.5]]
        uk = pa.UniversalKriging(x, y, v, variogram_model="exponential", variogram_parameters=[1.0, 0.5, 0.02],
                                 drift_terms=["regional_linear", "point_log"], point_drift=well)
        stu = ko.KrigingState(ndim=2, coords_orig=np.stack([x, y], 1), values=v, model="exponential",
                              params=ko.internal_parameters("exponential", [1.0, 0.5, 0.02]), regional_linear=True,
                              point_log=np.array(well))
        zu, su = uk.execute("points", pts[:, 0], pts[:, 1], backend="loop")
        zru, sru = ko.solve_points(stu, ko.adjust_for_anisotropy(pts, stu.center, stu.scaling, stu.angle))
        np.testing.assert_allclose(zu, zru, rtol=0, atol=Z_TOL)
        np.testing.assert_allclose(su, sru, rtol=0, atol=SS_TOL)


@pytest.mark.parametrize("case", ["clustered2d", "line2d", "outside2d", "large2d", "clustered3d", "flat3d", "k_eq_n"])
def test_moving_window_cell_grid_against_kdtree(case):
    """The neighbour search walks rings of station cells and stops at tau <= (ring * cell)^2; this pins it against the
    oracle's cKDTree query (ok.py:957-960) where that logic is stressed: very uneven station density, stations on a line
    (one grid dimension degenerate), points far outside the stations' bounding box, many cells, 3-D, a flat 3-D cloud,
    and a window that holds every station.  The local systems are computed from coordinates (no N x N matrix), the
    oracle cuts them out of the full matrix as the reference does."""
    import pykrige_amd as pa

    rng = np.random.default_rng({"clustered2d": 11, "line2d": 12, "outside2d": 13, "large2d": 14, "clustered3d": 15, "flat3d": 16, "k_eq_n": 17}[case])
    ndim, model, user = 2, "exponential", [1.0, 0.4, 0.02]
    if case == "clustered2d":
        n, k = 1500, 12
        c = np.concatenate([0.02 * rng.standard_normal((1300, 2)) + [0.2, 0.7], rng.random((200, 2))])
        pts = np.concatenate([rng.random((300, 2)), 0.05 * rng.standard_normal((100, 2)) + [0.2, 0.7]])
    elif case == "line2d":
        n, k = 600, 9
        c = np.stack([rng.random(n), np.full(n, 0.5)], 1)
        pts = rng.random((200, 2))
    elif case == "outside2d":
        n, k = 900, 20
        c = rng.random((n, 2))
        pts = np.concatenate([rng.random((50, 2)) * 40 - 20, [[-5.0, 0.5], [0.5, 9.0], [100.0, 100.0], [0.0, 0.0], [1.0, 1.0]]])
    elif case == "large2d":
        n, k = 8000, 16
        c = rng.random((n, 2))
        pts = rng.random((400, 2))
    elif case == "clustered3d":
        ndim, n, k, model, user = 3, 2500, 14, "spherical", [1.0, 0.6, 0.05]
        c = np.concatenate([0.03 * rng.standard_normal((2000, 3)) + [0.5, 0.5, 0.2], rng.random((500, 3))])
        pts = rng.random((300, 3))
    elif case == "flat3d":
        ndim, n, k, model, user = 3, 800, 10, "gaussian", [1.0, 0.5, 0.05]
        c = np.concatenate([rng.random((n, 2)), np.full((n, 1), 0.25)], 1)
        pts = rng.random((200, 3))
    else:  # k_eq_n
        n, k = 150, 150
        c = rng.random((n, 2))
        pts = rng.random((64, 2))
    n = c.shape[0]
    v = np.sin(5 * c[:, 0]) + np.cos(3 * c[:, 1]) + 0.1 * rng.standard_normal(n)
    st = ko.KrigingState(ndim=ndim, coords_orig=c, values=v, model=model, params=ko.internal_parameters(model, user),
                         scaling=[1.0] * (ndim - 1), angle=[0.0] * (2 * ndim - 3))
    zr, sr = ko.solve_points_moving_window(st, ko.adjust_for_anisotropy(pts.copy(), st.center, st.scaling, st.angle), k)
    if ndim == 2:
        m = pa.OrdinaryKriging(c[:, 0], c[:, 1], v, variogram_model=model, variogram_parameters=user)
        z, ss = m.execute("points", pts[:, 0], pts[:, 1], backend="loop", n_closest_points=k)
    else:
        m = pa.OrdinaryKriging3D(c[:, 0], c[:, 1], c[:, 2], v, variogram_model=model, variogram_parameters=user)
        z, ss = m.execute("points", pts[:, 0], pts[:, 1], pts[:, 2], backend="loop", n_closest_points=k)
    # BASELINE's 1e-8 on z holds for every point inside the stations' bounding box; the 1e-7 allowance exists for the points
    # placed far outside it (clustered / collinear cases), whose local systems are ill-conditioned -- printed when it is used
    dz = np.abs(z - zr)
    inside = np.all((pts >= c.min(axis=0)) & (pts <= c.max(axis=0)), axis=1)
    used = int((dz > 1e-8).sum())
    print("moving-window %s: worst |dz| %.2e (inside the bounding box %.2e); %d of %d points above 1e-8" % (
        case, dz.max(), dz[inside].max() if inside.any() else 0.0, used, dz.size))
    np.testing.assert_allclose(z, zr, rtol=0, atol=1e-7)
    np.testing.assert_allclose(ss, sr, rtol=0, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("ndim", [2, 3])
def test_moving_window_over_a_shuffled_point_list_is_sorted_on_the_device(ndim):
    """Small windows over points in no spatial order (the sklearn-side default n_closest_points = 10 on a scattered list): the
    library puts the points in Hilbert-curve order on the device (k_ps_*, the sorter of the range-aware contraction), runs the
    lane-per-point neighbour search on compact wavefronts and scatters z / sigma^2 back.  A point's result does not depend on its
    neighbours in the list: bit-identical to the unsorted run, and the oracle's on a sample."""
    import pykrige_amd as pa

    rng = np.random.default_rng(31)
    n, npts, k = 3000, 40000, 10
    c = rng.random((n, ndim))
    v = np.sin(5 * c[:, 0]) + np.cos(3 * c[:, 1]) + 0.1 * rng.standard_normal(n)
    pts = rng.random((npts, ndim))
    pts[:4] = c[:4]  # exact hits
    cls = pa.OrdinaryKriging if ndim == 2 else pa.OrdinaryKriging3D
    outs = []
    for sort in (1, 0):
        m = cls(*[c[:, d] for d in range(ndim)], v, variogram_model="exponential", variogram_parameters=[1.0, 0.4, 0.02])
        m._get_handle().set_option("sort_points", sort)
        z, ss = m.execute("points", *[pts[:, d] for d in range(ndim)], backend="loop", n_closest_points=k)
        assert m.last_timing["points_sorted"] == sort, m.last_timing
        outs.append((np.asarray(z).copy(), np.asarray(ss).copy()))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    st = ko.KrigingState(ndim=ndim, coords_orig=c, values=v, model="exponential", params=ko.internal_parameters("exponential", [1.0, 0.4, 0.02]),
                         scaling=[1.0] * (ndim - 1), angle=[0.0] * (2 * ndim - 3))
    sel = np.concatenate([np.arange(8), rng.choice(npts, 300, replace=False)])
    zr, sr = ko.solve_points_moving_window(st, ko.adjust_for_anisotropy(pts[sel].copy(), st.center, st.scaling, st.angle), k)
    np.testing.assert_allclose(outs[0][0][sel], zr, rtol=0, atol=1e-8)
    np.testing.assert_allclose(outs[0][1][sel], sr, rtol=0, atol=1e-6)


@pytest.mark.parametrize("k", [24, 40, 56, 72, 84, 92, 100, 104, 110, 124, 140, 156, 170, 188, 204, 220, 250])
def test_moving_window_every_register_class_against_the_oracle(k):
    """One window size per class {G, RI} of the LDL^T kernel (mikrige.hip::dispatch_mw_chol): {8,4} .. {8,13} on one wavefront,
    {16,7} .. {16,14} on 256 threads, {32,8} on 1024 -- the classes whose register allocation round 3 pinned (launch bounds, lean
    update) among them -- against the oracle's cKDTree + dense solve of the same windows (ok.py:929-986)."""
    import pykrige_amd as pa

    rng = np.random.default_rng(1000 + k)
    n = 320
    c = rng.random((n, 2))
    v = np.sin(5 * c[:, 0]) + np.cos(3 * c[:, 1]) + 0.1 * rng.standard_normal(n)
    pts = rng.random((48, 2))
    pts[:4] = c[:4]  # exact hits
    model, user = ("spherical", [1.0, 0.7, 0.05]) if k % 8 else ("exponential", [1.0, 0.5, 0.02])
    st = ko.KrigingState(ndim=2, coords_orig=c, values=v, model=model, params=ko.internal_parameters(model, user), scaling=[1.0], angle=[0.0])
    zr, sr = ko.solve_points_moving_window(st, ko.adjust_for_anisotropy(pts.copy(), st.center, st.scaling, st.angle), k)
    m = pa.OrdinaryKriging(c[:, 0], c[:, 1], v, variogram_model=model, variogram_parameters=user)
    z, ss = m.execute("points", pts[:, 0], pts[:, 1], backend="loop", n_closest_points=k)
    np.testing.assert_allclose(z, zr, rtol=0, atol=Z_TOL)
    np.testing.assert_allclose(ss, sr, rtol=0, atol=SS_TOL)


def test_moving_window_many_stations_without_the_full_matrix():
    """300 000 stations: the reference's moving window would first build a 720 GB kriging matrix (ok.py:975); here the
    neighbour search runs on the cell grid and each point's system comes from coordinates.  Checked on 256 points against
    cKDTree + a dense solve of the same (k+1) x (k+1) systems written out in NumPy."""
    import scipy.linalg
    from scipy.spatial import cKDTree

    import pykrige_amd as pa

    rng = np.random.default_rng(2025)
    n, k = 300000, 12
    x, y = rng.random(n), rng.random(n)
    v = np.sin(9 * x) * np.cos(7 * y) + 0.05 * rng.standard_normal(n)
    user = [1.0, 0.05, 0.01]
    ok = pa.OrdinaryKriging(x, y, v, variogram_model="exponential", variogram_parameters=user)
    px, py = rng.random(20000), rng.random(20000)
    z, ss = ok.execute("points", px, py, backend="loop", n_closest_points=k)
    par = ko.internal_parameters("exponential", user)
    d, idx = cKDTree(np.stack([x, y], 1)).query(np.stack([px[:256], py[:256]], 1), k=k)
    for i in range(256):
        sel = idx[i]
        c = np.stack([x[sel], y[sel]], 1)
        a = np.zeros((k + 1, k + 1))
        a[:k, :k] = -ko.variogram("exponential", par, np.linalg.norm(c[:, None] - c[None], axis=2))
        np.fill_diagonal(a, 0.0)
        a[k, :k] = a[:k, k] = 1.0
        b = np.append(-ko.variogram("exponential", par, d[i]), 1.0)
        w = scipy.linalg.solve(a, b)
        assert abs(z[i] - w[:k] @ v[sel]) <= Z_TOL and abs(ss[i] + w @ b) <= SS_TOL


def test_integration_md_stub_runs_as_written():
    """The ctypes stub INTEGRATION.md shows a PyKrige maintainer (section B) is executed here verbatim -- only the
    library path is made absolute -- on an object carrying the attributes OrdinaryKriging has at that point of execute()."""
    import os
    import re
    import types

    lib = _lib()
    text = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "INTEGRATION.md")).read()
    block = next(b for b in re.findall(r"```python\n(.*?)```", text, flags=re.S) if "def _hip_exec" in b)
    ns = {}
    exec(block.replace('C.CDLL("libmikrige.so")', "C.CDLL(%r)" % lib.LIB_PATH), ns)
    g = fx.load("ok2d_spherical_exact")
    st = fx.state_from("ok2d_spherical_exact", g)

    def spherical_variogram_model(m, d):  # only its __name__ is read, as by lib/variogram_models.pyx
        return None

    assert st.model == "spherical"
    fake = types.SimpleNamespace(X_ADJUSTED=st.coords_adj[:, 0], Y_ADJUSTED=st.coords_adj[:, 1], Z=st.values,
                                 variogram_function=spherical_variogram_model, variogram_model_parameters=list(st.params),
                                 eps=1e-10, exact_values=st.exact_values, coordinates_type="euclidean", pseudo_inv=False,
                                 pseudo_inv_type="pinv")
    rng = np.random.default_rng(3)
    pts = ko.adjust_for_anisotropy(rng.random((500, 2)), st.center, st.scaling, st.angle)
    mask = rng.random(500) < 0.2
    z, ss = ns["_hip_exec"](fake, pts[:, 0], pts[:, 1], mask)
    zr, sr = ko.solve_points(st, pts[~mask])
    np.testing.assert_allclose(z[~mask], zr, rtol=0, atol=Z_TOL)
    np.testing.assert_allclose(ss[~mask], sr, rtol=0, atol=SS_TOL)
    assert np.all(z[mask] == 0.0) and np.all(ss[mask] == 0.0)


def test_integration_md_grid_stub_runs_as_written():
    """The second stub of INTEGRATION.md section B -- style='grid' / 'masked' through mik_set_grid, the axes instead of the meshgrid
    -- executed verbatim (after the first stub, which it extends) on an object carrying OrdinaryKriging's attributes, against the
    reference's stored answer for an ANISOTROPIC fixture with exact hits."""
    import os
    import re
    import types

    lib = _lib()
    text = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    first = next(b for b in blocks if "def _hip_exec(" in b)
    second = next(b for b in blocks if "def _hip_exec_grid" in b)
    ns = {}
    exec(first.replace('C.CDLL("libmikrige.so")', "C.CDLL(%r)" % lib.LIB_PATH), ns)
    exec(second, ns)
    name = "ok2d_exponential_exact"
    g = fx.load(name)
    st = fx.state_from(name, g)

    def exponential_variogram_model(m, d):  # only its __name__ is read
        return None

    fake = types.SimpleNamespace(X_ADJUSTED=st.coords_adj[:, 0], Y_ADJUSTED=st.coords_adj[:, 1], Z=st.values,
                                 variogram_function=exponential_variogram_model, variogram_model_parameters=list(st.params),
                                 eps=1e-10, exact_values=st.exact_values, anisotropy_scaling=float(st.scaling[0]),
                                 anisotropy_angle=float(st.angle[0]), XCENTER=float(st.center[0]), YCENTER=float(st.center[1]))
    z, ss = ns["_hip_exec_grid"](fake, g["gridx"], g["gridy"])
    assert float(st.scaling[0]) != 1.0 or float(st.angle[0]) != 0.0  # the fixture really is anisotropic
    np.testing.assert_allclose(z, g["z"], rtol=0, atol=Z_TOL)
    np.testing.assert_allclose(ss, g["ss"], rtol=0, atol=SS_TOL)
    rng = np.random.default_rng(5)
    mask = rng.random(z.shape) < 0.3
    zm, sm = ns["_hip_exec_grid"](fake, g["gridx"], g["gridy"], mask)
    np.testing.assert_allclose(zm[~mask], g["z"][~mask], rtol=0, atol=Z_TOL)
    assert np.all(zm[mask] == 0.0) and np.all(sm[mask] == 0.0)


def test_pseudo_inverse_fast_path_refuses_what_it_cannot_prove():
    """pseudo_inv on a rank deficiency that is NOT duplicated stations -- collinear stations under a regional-linear drift make
    two drift columns linearly dependent -- : the null space is found numerically and deflated (factor_path 6, round 3), the
    result is scipy.linalg.pinv's; with the fast paths switched off the one-sided Jacobi pseudo-inverse (factor_path 4) gives
    the same matrix; and a regular matrix comes back as its plain inverse (factor_path 5)."""
    import scipy.linalg

    lib = _lib()
    rng = np.random.default_rng(77)
    n = 60
    x = rng.random(n)
    y = 2.0 * x + 0.25  # every station on one line: the x and y drift columns are dependent (rank M - 1), no duplicates
    v = np.sin(4 * x) + 0.1 * rng.standard_normal(n)
    st = ko.KrigingState(ndim=2, coords_orig=np.stack([x, y], 1), values=v, model="exponential",
                         params=ko.internal_parameters("exponential", [1.0, 0.5, 0.05]), regional_linear=True)
    a = ko.kriging_matrix(st)
    assert np.linalg.matrix_rank(a) == a.shape[0] - 1
    h = lib.Handle(0)
    h.set_problem(ndim=2, xs=st.coords_adj[:, 0], ys=st.coords_adj[:, 1], zs=None, values=v, model_id=lib.MODEL_IDS["exponential"],
                  params=st.params, regional_linear=True, pseudo_inv=1)
    h.factor()
    t = h.timing()
    assert t["factor_path"] == 6 and t["null_dim"] == 1, t
    pinv = scipy.linalg.pinv(a)
    fast = h.get_matrix(1)
    assert np.abs(fast - pinv).max() <= 1e-8 * np.abs(pinv).max()
    h.set_option("pinv_fast", 0)
    h.factor()
    assert h.timing()["factor_path"] == 4
    assert np.abs(h.get_matrix(1) - pinv).max() <= 1e-8 * np.abs(pinv).max()
    h.set_option("pinv_fast", 1)
    # larger cases of the same kind: (b) well-conditioned range (nugget) -> the deflated path; (c) duplicated stations on top and a
    # zero nugget (null space 1 + 3, range of condition 5e7): whichever path can PROVE its result -- the deflated inverse is
    # checked on the directions of the smallest non-zero eigenvalues and refused at |A X u - u| > 1e-7
    for tag, nug, dups in (("b", 0.02, 0), ("c", 0.0, 3)):
        n2 = 700
        x2 = rng.random(n2)
        if dups:
            x2[-dups:] = x2[:dups]
        y2l = 0.5 * x2 - 0.1
        v2 = np.cos(3 * x2) + 0.1 * rng.standard_normal(n2)
        stb = ko.KrigingState(ndim=2, coords_orig=np.stack([x2, y2l], 1), values=v2, model="exponential",
                              params=ko.internal_parameters("exponential", [1.0, 0.5, nug]), regional_linear=True)
        ab = ko.kriging_matrix(stb)
        h.set_problem(ndim=2, xs=stb.coords_adj[:, 0], ys=stb.coords_adj[:, 1], zs=None, values=v2, model_id=lib.MODEL_IDS["exponential"],
                      params=stb.params, regional_linear=True, pseudo_inv=1)
        h.factor()
        t = h.timing()
        pinvb = scipy.linalg.pinv(ab)
        err = np.abs(h.get_matrix(1) - pinvb).max() / np.abs(pinvb).max()
        print("collinear (%s), M = %d: factor_path %d, null_dim %d, invert %.1f ms, max|X - pinv| / max|pinv| = %.1e"
              % (tag, ab.shape[0], t["factor_path"], t["null_dim"], t["invert_ms"], err))
        assert err <= (1e-8 if tag == "b" else 1e-6)  # (c): a range of condition ~1e8 -- SciPy's own SVD is no better than 1e-8 there
        if tag == "b":
            assert t["factor_path"] == 6 and t["null_dim"] == 1
        else:
            assert t["factor_path"] in (4, 6)
    # regular matrix, pseudo_inv requested: the plain inverse, verified
    y2 = rng.random(n)
    st2 = ko.KrigingState(ndim=2, coords_orig=np.stack([x, y2], 1), values=v, model="exponential",
                          params=ko.internal_parameters("exponential", [1.0, 0.5, 0.05]))
    h.set_problem(ndim=2, xs=st2.coords_adj[:, 0], ys=st2.coords_adj[:, 1], zs=None, values=v, model_id=lib.MODEL_IDS["exponential"],
                  params=st2.params, pseudo_inv=1)
    h.factor()
    assert h.timing()["factor_path"] == 5
    inv = scipy.linalg.inv(ko.kriging_matrix(st2))
    assert np.abs(h.get_matrix(1) - inv).max() <= 1e-9 * np.abs(inv).max()
    h.close()
