"""GPU tests of behaviours the reference's own suite pins with known answers or equivalences
(/root/reference/tests/test_core.py; line ranges cited per test), restated against pykrige_amd.
Small systems (3-50 stations): they exercise the padded-matrix / single-tile corners of the device path."""
import numpy as np
import pytest
from pytest import approx

pytestmark = pytest.mark.gpu


def _pa():
    import pykrige_amd as pa

    return pa


def test_exact_interpolation_on_a_diagonal_of_three_stations():
    """test_force_exact (test_core.py:1510-1834): linear [1, 1]; z == datum and sigma^2 == 0 exactly at stations."""
    pa = _pa()
    d = np.array([[1.0, 1.0, 2.0], [2.0, 2.0, 1.5], [3.0, 3.0, 1.0]])
    ok = pa.OrdinaryKriging(d[:, 0], d[:, 1], d[:, 2], variogram_model="linear", variogram_parameters=[1.0, 1.0])
    for backend in ("vectorized", "loop", "C"):
        z, ss = ok.execute("grid", [1.0, 2.0, 3.0], [1.0, 2.0, 3.0], backend=backend)
        for k in range(3):
            assert z[k, k] == approx(d[k, 2]) and ss[k, k] == approx(0.0, abs=1e-12)
        assert ss[0, 2] != approx(0.0) and ss[2, 0] != approx(0.0)
        z, ss = ok.execute("points", [1.0, 2.0, 3.0, 3.0], [2.0, 1.0, 1.0, 3.0], backend=backend)
        assert all(ss[k] != approx(0.0) for k in range(3))
        assert z[3] == approx(1.0) and ss[3] == approx(0.0, abs=1e-12)
        z, ss = ok.execute("grid", np.arange(0.0, 4.0, 0.1), np.arange(0.0, 4.0, 0.1), backend=backend)
        for k, node in enumerate((10, 20, 30)):
            assert z[node, node] == approx(d[k, 2]) and ss[node, node] == approx(0.0, abs=1e-12)
        for a, b in ((0, 0), (15, 15), (10, 0), (0, 10), (20, 10), (10, 20), (30, 20), (20, 30)):
            assert ss[a, b] != approx(0.0)
    z, ss = ok.execute("grid", np.arange(0.0, 3.1, 0.1), np.arange(2.1, 3.1, 0.1), backend="vectorized")
    assert np.any(np.isclose(ss, 0)) and not np.any(np.isclose(ss[:9, :30], 0)) and not np.allclose(z[:9, :30], 0.0)
    z, ss = ok.execute("grid", np.arange(0.0, 1.9, 0.1), np.arange(2.1, 3.1, 0.1), backend="vectorized")
    assert not np.any(np.isclose(ss, 0))
    gx, gy = np.arange(2.5, 3.5, 0.1), np.arange(2.5, 3.5, 0.25)
    z, ss = ok.execute("masked", gx, gy, backend="vectorized", mask=np.asarray(np.meshgrid(gx, gy)[0] == 0.0))
    assert np.isclose(ss[2, 5], 0) and not np.allclose(ss, 0.0)


def test_exact_interpolation_3d():
    """test_force_exact_3d (test_core.py:2505-2560)."""
    pa = _pa()
    d = np.array([[1.0, 1.0, 1.0, 2.0], [2.0, 2.0, 2.0, 1.5], [3.0, 3.0, 3.0, 1.0]])
    k3 = pa.OrdinaryKriging3D(d[:, 0], d[:, 1], d[:, 2], d[:, 3], variogram_model="linear", variogram_parameters=[1.0, 1.0])
    g = [1.0, 2.0, 3.0]
    for backend in ("vectorized", "loop"):
        k, ss = k3.execute("grid", g, g, g, backend=backend)
        for q in range(3):
            assert k[q, q, q] == approx(d[q, 3]) and ss[q, q, q] == approx(0.0, abs=1e-12)
        assert ss[2, 0, 0] != approx(0.0) and ss[0, 2, 0] != approx(0.0)
        k, ss = k3.execute("points", [1.0, 2.0, 3.0, 3.0], [2.0, 1.0, 1.0, 3.0], [1.0, 1.0, 3.0, 3.0], backend=backend)
        assert k[3] == approx(1.0) and ss[3] == approx(0.0, abs=1e-12) and ss[0] != approx(0.0)
        ax = np.arange(0.0, 4.0, 0.5)
        k, ss = k3.execute("grid", ax, ax, ax, backend=backend)
        assert k.shape == (8, 8, 8) and k[2, 2, 2] == approx(2.0) and ss[4, 4, 4] == approx(0.0, abs=1e-12)


def test_exact_values_false_changes_only_the_station_nodes():
    """test_non_exact (test_core.py:430-487)."""
    pa = _pa()
    d = np.array([[0.0, 0.0, 0.47], [1.5, 1.5, 0.56], [3, 3, 0.74], [4.5, 4.5, 1.47]])
    g = np.arange(0.0, 4.51, 1.5)
    kw = dict(variogram_model="exponential", variogram_parameters=[500.0, 3000.0, 5.0])
    z, _ = pa.OrdinaryKriging(d[:, 0], d[:, 1], d[:, 2], **kw).execute("grid", g, g, backend="loop")
    zn, _ = pa.OrdinaryKriging(d[:, 0], d[:, 1], d[:, 2], exact_values=False, **kw).execute("grid", g, g, backend="loop")
    np.testing.assert_allclose(np.diag(z), d[:, 2])
    assert not np.allclose(np.diag(zn), d[:, 2])
    np.fill_diagonal(z, 0.0)
    np.fill_diagonal(zn, 0.0)
    np.testing.assert_allclose(z, zn, rtol=1e-7)


def test_ucla_universal_kriging_point():
    """test_uk_execute_single_point (test_core.py:856-895): lecture-note answer z = 567.54, sigma^2 = 9.044 (rel 0.1)."""
    pa = _pa()
    d = np.array([[61.0, 139.0, 477.0], [63.0, 140.0, 696.0], [64.0, 129.0, 227.0], [68.0, 128.0, 646.0],
                  [71.0, 140.0, 606.0], [73.0, 141.0, 791.0], [75.0, 128.0, 783.0]])
    uk = pa.UniversalKriging(d[:, 0], d[:, 1], d[:, 2], variogram_model="exponential",
                             variogram_parameters=[10.0, 9.99, 0.0], drift_terms=["regional_linear"])
    for backend in ("vectorized", "loop"):
        z, ss = uk.execute("points", np.array([65.0]), np.array([137.0]), backend=backend)
        assert 567.54 == approx(z[0], rel=0.1) and 9.044 == approx(ss[0], rel=0.1)
        z, ss = uk.execute("points", np.array([61.0]), np.array([139.0]), backend=backend)
        assert z[0] == approx(477.0, rel=1e-3) and ss[0] == approx(0.0, abs=1e-9)


def test_kitanidis_example_through_execute():
    """test_core_krige (test_core.py:378-427), Kitanidis ex. 3.2: z = 1.6364, sigma^2 = 0.4201."""
    pa = _pa()
    d = np.array([[9.7, 47.6, 1.22], [43.8, 24.6, 2.822]])
    ok = pa.OrdinaryKriging(d[:, 0], d[:, 1], d[:, 2], variogram_model="linear", variogram_parameters=[0.006, 0.1])
    z, ss = ok.execute("points", [18.8, 43.8], [67.9, 24.6], backend="loop")
    assert z[0] == approx(1.6364, rel=1e-4) and ss[0] == approx(0.4201, rel=1e-4)
    assert z[1] == approx(2.822, rel=1e-3) and ss[1] == approx(0.0, abs=1e-12)
    k3 = pa.OrdinaryKriging3D(d[:, 0], d[:, 1], np.ones(2), d[:, 2], variogram_model="linear", variogram_parameters=[0.006, 0.1])
    z, ss = k3.execute("points", [18.8], [67.9], [1.0], backend="loop")
    assert z[0] == approx(1.6364, rel=1e-4) and ss[0] == approx(0.4201, rel=1e-4)


@pytest.fixture
def sample():  # the reference's sample_data_2d (test_core.py:45-61)
    data = np.array([[0.3, 1.2, 0.47], [1.9, 0.6, 0.56], [1.1, 3.2, 0.74], [3.3, 4.4, 1.47], [4.7, 3.8, 1.74]])
    return data, np.arange(0.0, 6.0, 1.0), np.arange(0.0, 5.5, 0.5)


def test_drift_constructions_are_equivalent(sample):
    """test_ok_uk_produce_same_result, test_uk_specified_drift, test_uk_functional_drift (test_core.py:1020-1066,
    1257-1476): UK without drift == OK; 'specified' [x, y] == 'functional' [x, y] == regional_linear; a well given as a
    specified drift == point_log (incl. the -inf -> -100 rule when a grid node sits on the well)."""
    pa = _pa()
    data, gx, gy = sample
    xg, yg = np.meshgrid(gx, gy)
    x, y, v = data[:, 0], data[:, 1], data[:, 2]
    kw = dict(variogram_model="linear", variogram_parameters=[1.0, 0.1])
    z0, s0 = pa.OrdinaryKriging(x, y, v, **kw).execute("grid", gx, gy, backend="loop")
    z1, s1 = pa.UniversalKriging(x, y, v, **kw).execute("grid", gx, gy, backend="loop")
    np.testing.assert_allclose(z0, z1, atol=1e-10)
    np.testing.assert_allclose(s0, s1, atol=1e-10)
    zl, sl = pa.UniversalKriging(x, y, v, drift_terms=["regional_linear"], **kw).execute("grid", gx, gy, backend="loop")
    uk_spec = pa.UniversalKriging(x, y, v, drift_terms=["specified"], specified_drift=[x, y], **kw)
    with pytest.raises(ValueError):
        uk_spec.execute("grid", gx, gy, specified_drift_arrays=[gx, gy])
    with pytest.raises(TypeError):
        uk_spec.execute("grid", gx, gy, specified_drift_arrays=gx)
    with pytest.raises(ValueError):
        uk_spec.execute("grid", gx, gy, specified_drift_arrays=[xg])
    zs, ss_ = uk_spec.execute("grid", gx, gy, specified_drift_arrays=[xg, yg], backend="loop")
    np.testing.assert_allclose(zs, zl, atol=1e-9)
    np.testing.assert_allclose(ss_, sl, atol=1e-9)
    zf, sf = pa.UniversalKriging(x, y, v, drift_terms=["functional"], functional_drift=[lambda a, b: a, lambda a, b: b],
                                 **kw).execute("grid", gx, gy, backend="loop")
    np.testing.assert_allclose(zf, zl, atol=1e-9)
    np.testing.assert_allclose(sf, sl, atol=1e-9)
    # a well ON a grid node: log distance -inf -> -100 (uk.py:892-895)
    well = np.array([[1.0, 1.0, -1.0]])
    with np.errstate(divide="ignore"):
        pl_grid = -well[0, 2] * np.log(np.sqrt((xg - well[0, 0]) ** 2 + (yg - well[0, 1]) ** 2))
        pl_data = -well[0, 2] * np.log(np.sqrt((x - well[0, 0]) ** 2 + (y - well[0, 1]) ** 2))
    pl_grid[np.isinf(pl_grid)] = -100.0 * well[0, 2] * -1.0
    zw, sw = pa.UniversalKriging(x, y, v, drift_terms=["point_log"], point_drift=well, **kw).execute("grid", gx, gy, backend="loop")
    zp, sp = pa.UniversalKriging(x, y, v, drift_terms=["specified"], specified_drift=[pl_data], **kw).execute(
        "grid", gx, gy, specified_drift_arrays=[pl_grid], backend="loop")
    np.testing.assert_allclose(zw, zp, atol=1e-9)
    np.testing.assert_allclose(sw, sp, atol=1e-9)
    with pytest.raises(ValueError):
        pa.UniversalKriging(x, y, v, drift_terms=["specified"], **dict(kw, variogram_parameters=[1.0, 0.1]), specified_drift=[])
    with pytest.raises(ValueError):
        pa.UniversalKriging(x, y, v, drift_terms=["specified"], specified_drift=[x[:2]], **kw)


def test_uk3d_drift_equivalences_and_shapes():
    """test_uk3d_specified_drift / test_uk3d_functional_drift / test_ok3d_uk3d_and_backends_produce_same_results
    (test_core.py:2563-2688, 2020-2094) and the (nz, ny, nx) output convention (ok3d.py:929-930)."""
    pa = _pa()
    rng = np.random.default_rng(12)
    x, y, zc, v = rng.random(40) * 4, rng.random(40) * 3, rng.random(40) * 2, rng.random(40)
    gx, gy, gz = np.linspace(0, 4, 6), np.linspace(0, 3, 5), np.linspace(0, 2, 4)
    kw = dict(variogram_model="exponential", variogram_parameters=[1.0, 2.0, 0.05])
    k0, s0 = pa.OrdinaryKriging3D(x, y, zc, v, **kw).execute("grid", gx, gy, gz, backend="loop")
    k1, s1 = pa.UniversalKriging3D(x, y, zc, v, **kw).execute("grid", gx, gy, gz, backend="vectorized")
    assert k0.shape == (4, 5, 6)
    np.testing.assert_allclose(k0, np.ma.getdata(k1), atol=1e-10)
    np.testing.assert_allclose(s0, np.ma.getdata(s1), atol=1e-10)
    kl, sl = pa.UniversalKriging3D(x, y, zc, v, drift_terms=["regional_linear"], **kw).execute("grid", gx, gy, gz, backend="loop")
    zg, yg, xg = np.meshgrid(gz, gy, gx, indexing="ij")
    ks, ss = pa.UniversalKriging3D(x, y, zc, v, drift_terms=["specified"], specified_drift=[x, y, zc], **kw).execute(
        "grid", gx, gy, gz, specified_drift_arrays=[xg, yg, zg], backend="loop")
    kf, sf = pa.UniversalKriging3D(x, y, zc, v, drift_terms=["functional"],
                                   functional_drift=[lambda a, b, c: a, lambda a, b, c: b, lambda a, b, c: c], **kw).execute(
        "grid", gx, gy, gz, backend="loop")
    for kk, s_ in ((ks, ss), (kf, sf)):
        np.testing.assert_allclose(kk, kl, atol=1e-9)
        np.testing.assert_allclose(s_, sl, atol=1e-9)
    mask = rng.random((4, 5, 6)) < 0.3
    km, sm = pa.OrdinaryKriging3D(x, y, zc, v, **kw).execute("masked", gx, gy, gz, mask=mask, backend="loop")
    kt, _ = pa.OrdinaryKriging3D(x, y, zc, v, **kw).execute("masked", gx, gy, gz, mask=mask.swapaxes(0, 2), backend="loop")
    assert km[mask].mask.all() and not km[~mask].mask.any()
    np.testing.assert_array_equal(np.ma.getdata(km), np.ma.getdata(kt))
    np.testing.assert_allclose(np.ma.getdata(km)[~mask], k0[~mask], atol=1e-12)
    with pytest.raises(ValueError):
        pa.OrdinaryKriging3D(x, y, zc, v, **kw).execute("masked", gx, gy, gz, mask=np.zeros((2, 2, 2), bool))
    with pytest.raises(ValueError):
        pa.OrdinaryKriging3D(x, y, zc, v, **kw).execute("masked", gx, gy, gz, mask=np.zeros((5, 6), bool))


def test_core_krige_and_find_statistics_function_twins():
    """core._krige (test_core.py:378-427, Kitanidis ex. 3.2) and core._find_statistics with the reference's signatures,
    variogram given as a named function."""
    from pykrige_amd import core, variogram_models
    from tests import _fixtures as fx

    d = np.array([[9.7, 47.6, 1.22], [43.8, 24.6, 2.822]])
    z, ss = core._krige(d[:, :2], d[:, 2], np.array([18.8, 67.9]), variogram_models.linear_variogram_model, [0.006, 0.1], "euclidean")
    assert z == approx(1.6364, rel=1e-4) and ss == approx(0.4201, rel=1e-4)
    z, ss = core._krige(d[:, :2], d[:, 2], np.array([43.8, 24.6]), variogram_models.linear_variogram_model, [0.006, 0.1], "euclidean")
    assert z == approx(2.822, rel=1e-3) and ss == approx(0.0, abs=1e-12)
    d3 = np.array([[9.7, 47.6, 1.0, 1.22], [43.8, 24.6, 1.0, 2.822]])
    z, ss = core._krige(d3[:, :3], d3[:, 3], np.array([18.8, 67.9, 1.0]), variogram_models.linear_variogram_model, [0.006, 0.1], "euclidean")
    assert z == approx(1.6364, rel=1e-4) and ss == approx(0.4201, rel=1e-4)
    g = fx.load("stats_find_statistics")
    delta, sigma, eps = core._find_statistics(np.stack([g["x"], g["y"]], 1), g["v"], variogram_models.exponential_variogram_model,
                                              [0.9, 0.3, 0.1], "euclidean")
    np.testing.assert_allclose(delta, g["delta_exp"], atol=1e-8)
    np.testing.assert_allclose(sigma, g["sigma_exp"], atol=1e-8)
    np.testing.assert_allclose([core.calcQ1(eps), core.calcQ2(eps), core.calc_cR(core.calcQ2(eps), sigma)], g["q_exp"], rtol=1e-7)
    # a callable that is not one of the six named functions is a custom variogram: evaluated on the host, same numbers
    k1, s1 = core._krige(d[:, :2], d[:, 2], np.array([1.0, 1.0]), variogram_models.linear_variogram_model, [1.0, 1.0], "euclidean")
    k2, s2 = core._krige(d[:, :2], d[:, 2], np.array([1.0, 1.0]), lambda m, x: m[0] * x + m[1], [1.0, 1.0], "euclidean")
    assert k2 == approx(k1, abs=1e-10) and s2 == approx(s1, abs=1e-10)
    # pseudo_inv: duplicated stations, least-squares solution (core.py:749-750) = device pseudo-inverse; mean of the redundant data
    dup = np.array([[0.0, 0.0, 1.0], [0.0, 0.0, 3.0], [1.0, 0.0, 6.0], [0.3, 0.8, 2.0]])
    kz, _ = core._krige(dup[:, :2], dup[:, 2], np.array([0.0, 0.0]), variogram_models.linear_variogram_model, [1.0, 0.0], "euclidean",
                        pseudo_inv=True)
    assert kz == approx(2.0, abs=1e-9)


def test_sklearn_side_callers_match_the_reference():
    """compat.Krige for the four methods, rk.RegressionKriging and ck.ClassificationKriging against what the REAL reference
    returned for the same calls (tests/golden/sk_callers.npz <- oracle/make_golden_extra.py --sk; reference tests:
    test_api.py:15-30, test_regression_krige.py:33-64, test_classification_krige.py:30-64).  Krige.execute defaults to
    backend="loop" with n_closest_points, i.e. the device moving-window path for the ordinary methods."""
    pytest.importorskip("sklearn")
    from sklearn.linear_model import LinearRegression, LogisticRegression

    from tests import _fixtures as fx
    from pykrige_amd.ck import ClassificationKriging
    from pykrige_amd.compat import Krige
    from pykrige_amd.rk import RegressionKriging

    g = fx.load("sk_callers")
    X3, y, Q3 = g["X3"], g["y"], g["Q3"]
    cases = {"ordinary": dict(variogram_model="exponential", n_closest_points=8, nlags=6),
             "universal": dict(variogram_model="linear", drift_terms=["regional_linear"]),
             "ordinary3d": dict(variogram_model="spherical", n_closest_points=12, anisotropy_scaling=(1.5, 0.7)),
             "universal3d": dict(variogram_model="gaussian", variogram_parameters=[1.0, 0.8, 0.05],
                                 drift_terms=["regional_linear"])}
    for method, kw in cases.items():
        d = 3 if method.endswith("3d") else 2
        k = Krige(method=method, **kw)
        k.fit(X3[:, :d].copy(), y)
        np.testing.assert_allclose(k.model.variogram_model_parameters, g["krige_%s_par" % method], rtol=1e-6, atol=1e-9)
        pred, var = k.execute(k._dimensionality_check(Q3[:, :d].copy(), ext="points"))
        # the fitted variogram parameters agree to ~1e-7 relative (iterative fit), so the kriged values do to ~1e-6
        np.testing.assert_allclose(pred, g["krige_%s_pred" % method], rtol=0, atol=2e-6, err_msg=method)
        np.testing.assert_allclose(var, g["krige_%s_var" % method], rtol=0, atol=2e-6, err_msg=method)
        np.testing.assert_allclose(k.predict(Q3[:, :d].copy()), pred, rtol=0, atol=0)
    with pytest.raises(ValueError):
        Krige(method="simple")
    with pytest.raises(ValueError):
        Krige(method="ordinary3d").fit(X3[:, :2], y)
    rk = RegressionKriging(regression_model=LinearRegression(), method="ordinary", variogram_model="spherical", n_closest_points=10)
    rk.fit(g["P"], X3[:, :2].copy(), y)
    np.testing.assert_allclose(rk.predict(g["PQ"], Q3[:, :2].copy()), g["rk_pred"], rtol=0, atol=2e-6)
    assert rk.score(g["P"][:40], X3[:40, :2] + 0.01, y[:40]) == approx(float(g["rk_score"]), abs=1e-6)
    with pytest.raises(RuntimeError):
        RegressionKriging(regression_model=LogisticRegression())
    from sklearn.naive_bayes import GaussianNB  # closed-form learner: identical probabilities on every host

    ck = ClassificationKriging(classification_model=GaussianNB(), method="ordinary",
                               variogram_model="exponential", variogram_parameters=[450.0, 0.4, 5.0], n_closest_points=10)
    ck.fit(g["P"], X3[:, :2].copy(), g["labels"])
    got_par = np.array([k.model.variogram_model_parameters for k in ck.krige])
    np.testing.assert_allclose(got_par, g["ck_par"], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(ck.krige_residual(Q3[:, :2].copy()), g["ck_residual"], rtol=0, atol=5e-6)
    assert np.array_equal(ck.predict(g["PQ"], Q3[:, :2].copy()), g["ck_pred"])


def test_get_kriging_matrix_method():
    """ok._get_kriging_matrix(n) / uk._get_kriging_matrix(n) (ok.py:626-648, uk.py:861-920): the device-assembled matrix
    copied back equals the oracle's restatement, drift columns and borders included."""
    from oracle import kriging_oracle as ko
    from tests import _fixtures as fx

    pa = _pa()
    (x, y), v = fx.synth(31, 60, 2)
    ok = pa.OrdinaryKriging(x, y, v, variogram_model="gaussian", variogram_parameters=[1.0, 0.4, 0.05], anisotropy_scaling=1.7,
                            anisotropy_angle=25.0)
    st = ko.KrigingState(ndim=2, coords_orig=np.stack([x, y], 1), values=v, model="gaussian",
                         params=ko.internal_parameters("gaussian", [1.0, 0.4, 0.05]), scaling=[1.7], angle=[25.0])
    a = ok._get_kriging_matrix(60)
    assert a.shape == (61, 61)
    np.testing.assert_allclose(a, ko.kriging_matrix(st), rtol=0, atol=1e-13)
    wells = [[0.3, 0.6, 1.0]]
    uk = pa.UniversalKriging(x, y, v, variogram_model="linear", variogram_parameters=[1.2, 0.1],
                             drift_terms=["regional_linear", "point_log"], point_drift=wells)
    stu = ko.KrigingState(ndim=2, coords_orig=np.stack([x, y], 1), values=v, model="linear", params=[1.2, 0.1],
                          regional_linear=True, point_log=np.array(wells))
    au = uk._get_kriging_matrix(60)
    assert au.shape == (64, 64)
    np.testing.assert_allclose(au, ko.kriging_matrix(stu), rtol=0, atol=1e-12)


def test_custom_variogram_callable_matches_the_reference():
    """variogram_model='custom' (ok.py:247-254; test_core.py test_custom_variogram): the user's Python callable maps
    device-computed distances to semivariances on the host (mik_set_custom_variogram), everything else -- geometry,
    matrix borders and drifts, the eps rule, inverse, contraction, moving window, statistics -- runs on the device.
    Against the real reference with the same callable (tests/golden/custom_variogram.npz)."""
    from oracle.make_golden_extra import stable_variogram
    from tests import _fixtures as fx

    pa = _pa()
    g = fx.load("custom_variogram")
    par = [float(p) for p in g["par"]]
    ok = pa.OrdinaryKriging(g["x"], g["y"], g["v"], variogram_model="custom", variogram_parameters=par,
                            variogram_function=stable_variogram, anisotropy_scaling=1.4, anisotropy_angle=15.0, enable_statistics=True)
    z, ss = ok.execute("grid", g["gx"], g["gy"], backend="vectorized")
    np.testing.assert_allclose(z, g["ok_z"], rtol=0, atol=1e-8)
    np.testing.assert_allclose(ss, g["ok_ss"], rtol=0, atol=1e-6)
    assert ok.last_timing["factor_path"] == 2  # no sill to shift by: pivoted elimination
    zk, ssk = ok.execute("grid", g["gx"], g["gy"], backend="loop", n_closest_points=9)
    np.testing.assert_allclose(zk, g["ok_zk"], rtol=0, atol=1e-8)
    np.testing.assert_allclose(ssk, g["ok_ssk"], rtol=0, atol=1e-6)
    np.testing.assert_allclose([ok.Q1, ok.Q2, ok.cR], g["ok_q"], rtol=1e-7)
    np.testing.assert_allclose(ok.epsilon, g["ok_eps"], rtol=0, atol=1e-7)
    uk = pa.UniversalKriging(g["x"], g["y"], g["v"], variogram_model="custom", variogram_parameters=par,
                             variogram_function=stable_variogram, drift_terms=["regional_linear"])
    z, ss = uk.execute("grid", g["gx"], g["gy"], backend="loop")
    np.testing.assert_allclose(z, g["uk_z"], rtol=0, atol=1e-8)
    np.testing.assert_allclose(ss, g["uk_ss"], rtol=0, atol=1e-6)
    k3 = pa.OrdinaryKriging3D(g["x3"], g["y3"], g["z3"], g["v3"], variogram_model="custom", variogram_parameters=par,
                              variogram_function=stable_variogram)
    z, ss = k3.execute("grid", g["g3x"], g["g3y"], g["g3z"], backend="vectorized")
    np.testing.assert_allclose(z, g["k3_z"], rtol=0, atol=1e-8)
    np.testing.assert_allclose(ss, g["k3_ss"], rtol=0, atol=1e-6)
    lags, curve = ok.get_variogram_points()
    np.testing.assert_allclose(curve, stable_variogram(par, lags))
    # the multi-chunk route (every chunk's distances visit the host) and a model switch back to a named variogram
    ok._get_handle().set_option("chunk", 128)
    z2, _ = ok.execute("grid", g["gx"], g["gy"], backend="loop")
    np.testing.assert_allclose(z2, g["ok_z"], rtol=0, atol=1e-8)
    ok.update_variogram_model("exponential", [1.0, 0.4, 0.02])
    z3, _ = ok.execute("points", g["x"][:3], g["y"][:3], backend="loop")
    np.testing.assert_allclose(z3, g["v"][:3], rtol=0, atol=1e-8)
    with pytest.raises(ValueError):
        pa.OrdinaryKriging(g["x"], g["y"], g["v"], variogram_model="custom", variogram_parameters=par)  # no callable
    with pytest.raises(ValueError):
        pa.OrdinaryKriging(g["x"], g["y"], g["v"], variogram_model="custom", variogram_function=stable_variogram)  # no parameters
    with pytest.raises(TypeError):
        pa.OrdinaryKriging(g["x"], g["y"], g["v"], variogram_model="custom", variogram_function=stable_variogram,
                           variogram_parameters={"sill": 1.0})

    class FakeCovModel:  # the attributes a GSTools CovModel brings (ok.py:223-239); GSTools itself is not installed
        pykrige_kwargs = {}
        field_dim, latlon, pykrige_anis, pykrige_angle = 2, False, 1.4, 15.0

        @staticmethod
        def pykrige_vario(args=None, r=0):
            return stable_variogram(par, r)

    gs = pa.OrdinaryKriging(g["x"], g["y"], g["v"], variogram_model=FakeCovModel())
    z, ss = gs.execute("grid", g["gx"], g["gy"], backend="loop")
    np.testing.assert_allclose(z, g["ok_z"], rtol=0, atol=1e-8)
