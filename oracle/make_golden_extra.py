"""Second batch of golden vectors from the REAL reference (PyKrige 1.7.3 at /root/reference/src):
  ref_test_uk_external ... test_uk_with_external_drift (tests/test_core.py:1479-1507): DEM drift, KT3D-style answer grid
  uk2d_external_z ....... synthetic external_Z + regional_linear, off-node points (bilinear lookup parity)
  fit_<model> ........... constructor-time variogram fit (lags, semivariance, fitted parameters)
  pseudo_dup ............ duplicate stations with pseudo_inv (tests/test_core.py:2913-2949)
TEST INFRASTRUCTURE; run in the build container only:  python oracle/make_golden_extra.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import OUT, REF_DATA, _import_reference, synth  # noqa: E402


def main():
    _import_reference(False)
    import pykrige.kriging_tools as kt
    from pykrige.ok import OrdinaryKriging
    from pykrige.ok3d import OrdinaryKriging3D
    from pykrige.uk import UniversalKriging

    def arr(x):
        return np.ma.getdata(x).astype(np.float64)

    data = np.genfromtxt(os.path.join(REF_DATA, "test_data.txt"))
    dem, demx, demy, _, _ = kt.read_asc_grid(os.path.join(REF_DATA, "test3_dem.asc"))
    ans, gridx, gridy, _, _ = kt.read_asc_grid(os.path.join(REF_DATA, "test3_answer.asc"))
    uk = UniversalKriging(data[:, 0], data[:, 1], data[:, 2], variogram_model="spherical",
                          variogram_parameters=[500.0, 3000.0, 0.0], drift_terms=["external_Z"], external_drift=dem,
                          external_drift_x=demx, external_drift_y=demy)
    z, ss = uk.execute("grid", gridx, gridy, backend="vectorized")
    np.savez_compressed(os.path.join(OUT, "ref_test_uk_external.npz"), x=data[:, 0], y=data[:, 1], v=data[:, 2],
                        model="spherical", params_user=[500.0, 3000.0, 0.0], dem=dem, demx=demx, demy=demy,
                        gridx=gridx, gridy=gridy, z=arr(z), ss=arr(ss), answer=ans)
    # synthetic external_Z + regional_linear; kriging points off the DEM nodes
    (x, y), v = synth(77, 150, 2)
    ex, ey = np.linspace(-0.1, 1.1, 25), np.linspace(-0.2, 1.2, 31)
    EX, EY = np.meshgrid(ex, ey)
    edem = np.sin(2 * EX) + EY**2
    uk = UniversalKriging(x, y, v, variogram_model="exponential", variogram_parameters=[1.0, 0.3, 0.02],
                          drift_terms=["regional_linear", "external_Z"], external_drift=edem, external_drift_x=ex,
                          external_drift_y=ey)
    gx_, gy_ = np.linspace(0, 1, 17), np.linspace(0, 1, 13)
    z, ss = uk.execute("grid", gx_, gy_, backend="vectorized")
    np.savez_compressed(os.path.join(OUT, "uk2d_external_z.npz"), x=x, y=y, v=v, model="exponential",
                        params_user=[1.0, 0.3, 0.02], dem=edem, demx=ex, demy=ey, gridx=gx_, gridy=gy_, z=arr(z),
                        ss=arr(ss), regional_linear=True, z_scalars=uk.z_scalars)
    # variogram fits
    (x, y), v = synth(78, 200, 2)
    (x3, y3, z3), v3 = synth(79, 150, 3)
    fits = {}
    for model in ("linear", "power", "gaussian", "spherical", "exponential", "hole-effect"):
        for weight in (False, True):
            ok = OrdinaryKriging(x, y, v, variogram_model=model, nlags=8, weight=weight, anisotropy_scaling=2.0,
                                 anisotropy_angle=30.0)
            key = "%s_%d" % (model.replace("-", ""), int(weight))
            fits["lags_" + key], fits["semi_" + key] = ok.lags, ok.semivariance
            fits["par_" + key] = np.asarray(ok.variogram_model_parameters, dtype=np.float64)
    ok3 = OrdinaryKriging3D(x3, y3, z3, v3, variogram_model="spherical", nlags=6)
    fits["lags_3d"], fits["semi_3d"], fits["par_3d"] = ok3.lags, ok3.semivariance, np.asarray(ok3.variogram_model_parameters)
    np.savez_compressed(os.path.join(OUT, "fit_variograms.npz"), x=x, y=y, v=v, x3=x3, y3=y3, z3=z3, v3=v3, **fits)
    # pseudo-inverse with duplicated stations
    d = np.array([[0.0, 0.0, 1.0], [0.0, 0.0, 3.0], [1.0, 0.0, 6.0], [0.3, 0.8, 2.0]])
    out = {}
    for p_type in ("pinv", "pinvh"):
        ok = OrdinaryKriging(d[:, 0], d[:, 1], d[:, 2], variogram_parameters=[1.0, 0.0], pseudo_inv=True, pseudo_inv_type=p_type)
        z, ss = ok.execute("grid", np.linspace(0, 1, 5), np.linspace(0, 1, 4), backend="vectorized")
        out["z_" + p_type], out["ss_" + p_type] = arr(z), arr(ss)
    np.savez_compressed(os.path.join(OUT, "pseudo_dup.npz"), d=d, **out)
    print("wrote extra fixtures")


if __name__ == "__main__" and not {"--mw", "--geo", "--stats", "--sk", "--tools", "--custom", "--aniso"} & set(sys.argv):
    main()


def moving_window():
    """n_closest_points fixtures: OK2D backend='loop' and (if oracle/_ref is built) 'C'; OK3D backend='loop'."""
    import pykrige.lib

    with_c = os.path.isdir(os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "pykrige_lib"))
    if with_c:
        pykrige.lib.__path__.append(os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "pykrige_lib"))
    from pykrige.ok import OrdinaryKriging
    from pykrige.ok3d import OrdinaryKriging3D

    def arr(x):
        return np.ma.getdata(x).astype(np.float64)

    (x, y), v = synth(90, 400, 2)
    gx_, gy_ = np.linspace(0, 1, 21), np.linspace(0, 1, 17)
    x[:5], y[:5] = gx_[[3, 7, 11, 15, 19]], gy_[[2, 5, 8, 11, 14]]
    out = dict(x=x, y=y, v=v, model="spherical", params_user=[1.0, 0.4, 0.05], gridx=gx_, gridy=gy_, scaling=2.0, angle=20.0)
    ok = OrdinaryKriging(x, y, v, variogram_model="spherical", variogram_parameters=[1.0, 0.4, 0.05],
                         anisotropy_scaling=2.0, anisotropy_angle=20.0)
    rng = np.random.default_rng(91)
    mask = rng.random((17, 21)) < 0.25
    for k in (2, 10, 31, 70, 128, 200, 400):  # > 127: the HBM-resident device variant; 400 = every station
        z, ss = ok.execute("grid", gx_, gy_, backend="loop", n_closest_points=k)
        out["z_k%d" % k], out["ss_k%d" % k] = arr(z), arr(ss)
        if with_c:
            zc, ssc = ok.execute("grid", gx_, gy_, backend="C", n_closest_points=k)
            out["zc_k%d" % k], out["ssc_k%d" % k] = arr(zc), arr(ssc)
    zm, ssm = ok.execute("masked", gx_, gy_, mask=mask, backend="loop", n_closest_points=10)
    out.update(mask=mask, zm_k10=arr(zm), ssm_k10=arr(ssm))
    np.savez_compressed(os.path.join(OUT, "mw_ok2d.npz"), **out)
    (x, y, zc), v = synth(92, 300, 3)
    g3x, g3y, g3z = np.linspace(0, 1, 9), np.linspace(0, 1, 7), np.linspace(0, 1, 5)
    k3 = OrdinaryKriging3D(x, y, zc, v, variogram_model="exponential", variogram_parameters=[1.0, 0.5, 0.02],
                           anisotropy_scaling_y=1.5, anisotropy_scaling_z=2.0, anisotropy_angle_x=10.0,
                           anisotropy_angle_y=20.0, anisotropy_angle_z=30.0)
    out = dict(x=x, y=y, zc=zc, v=v, model="exponential", params_user=[1.0, 0.5, 0.02], gridx=g3x, gridy=g3y, gridz=g3z,
               scaling=[1.5, 2.0], angle=[10.0, 20.0, 30.0])
    for k in (8, 20, 150):
        z, ss = k3.execute("grid", g3x, g3y, g3z, backend="loop", n_closest_points=k)
        out["z_k%d" % k], out["ss_k%d" % k] = arr(z), arr(ss)
    np.savez_compressed(os.path.join(OUT, "mw_ok3d.npz"), **out)
    print("wrote moving-window fixtures (C backend: %s)" % with_c)


if __name__ == "__main__" and "--mw" in sys.argv:
    _import_reference(False)
    moving_window()


def geographic():
    """coordinates_type='geographic' (ok.py:634-640, 990-996; tests/test_core.py:2750-2881): full-matrix and moving-window."""
    import pykrige.lib

    refdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "pykrige_lib")
    with_c = os.path.isdir(refdir)
    if with_c and refdir not in list(pykrige.lib.__path__):
        pykrige.lib.__path__.append(refdir)
    from pykrige.ok import OrdinaryKriging

    def arr(x):
        return np.ma.getdata(x).astype(np.float64)

    rng = np.random.default_rng(95)
    n = 300
    lon = rng.uniform(-30.0, 50.0, n)
    lat = rng.uniform(-60.0, 75.0, n)
    v = np.sin(np.radians(lon) * 2) * np.cos(np.radians(lat) * 3) + 0.1 * rng.standard_normal(n)
    glon, glat = np.linspace(-30.0, 50.0, 19), np.linspace(-60.0, 75.0, 15)
    lon[:4], lat[:4] = glon[[2, 6, 10, 14]], glat[[1, 4, 8, 12]]
    out = dict(x=lon, y=lat, v=v, model="exponential", params_user=[1.0, 40.0, 0.05], gridx=glon, gridy=glat, geographic=True)
    ok = OrdinaryKriging(lon, lat, v, variogram_model="exponential", variogram_parameters=[1.0, 40.0, 0.05],
                         coordinates_type="geographic")
    z, ss = ok.execute("grid", glon, glat, backend="vectorized")
    out.update(z=arr(z), ss=arr(ss), A=ok._get_kriging_matrix(n))
    for k in (6, 20, 140):
        zk, ssk = ok.execute("grid", glon, glat, backend="loop", n_closest_points=k)
        out["z_k%d" % k], out["ss_k%d" % k] = arr(zk), arr(ssk)
        if with_c:
            zc, ssc = ok.execute("grid", glon, glat, backend="C", n_closest_points=k)
            out["zc_k%d" % k], out["ssc_k%d" % k] = arr(zc), arr(ssc)
    okf = OrdinaryKriging(lon, lat, v, variogram_model="spherical", coordinates_type="geographic", nlags=7)
    out.update(fit_lags=okf.lags, fit_semi=okf.semivariance, fit_par=np.asarray(okf.variogram_model_parameters))
    np.savez_compressed(os.path.join(OUT, "geo_ok2d.npz"), **out)
    print("wrote geographic fixture (C backend: %s)" % with_c)


if __name__ == "__main__" and "--geo" in sys.argv:
    _import_reference(False)
    geographic()


def statistics():
    """core._find_statistics (core.py:759-836) of the REAL reference: delta, sigma, epsilon, Q1, Q2, cR."""
    import pykrige.core as core
    import pykrige.variogram_models as vm

    out = {}
    (x, y), v = synth(96, 300, 2)
    X = np.stack([x, y], 1)
    for tag, fn, par in (("exp", vm.exponential_variogram_model, [0.9, 0.3, 0.1]), ("lin", vm.linear_variogram_model, [1.3, 0.05]),
                         ("sph", vm.spherical_variogram_model, [0.8, 0.5, 0.05])):
        d, s, e = core._find_statistics(X, v, fn, par, "euclidean")
        out.update({"delta_" + tag: d, "sigma_" + tag: s, "eps_" + tag: e,
                    "q_" + tag: np.array([core.calcQ1(e), core.calcQ2(e), core.calc_cR(core.calcQ2(e), s)])})
    (x3, y3, z3), v3 = synth(97, 200, 3)
    d, s, e = core._find_statistics(np.stack([x3, y3, z3], 1), v3, vm.gaussian_variogram_model, [0.9, 0.5, 0.1], "euclidean")
    out.update(delta_3d=d, sigma_3d=s, eps_3d=e, q_3d=np.array([core.calcQ1(e), core.calcQ2(e), core.calc_cR(core.calcQ2(e), s)]))
    rng = np.random.default_rng(98)
    lon, lat = rng.uniform(-20, 40, 150), rng.uniform(-50, 60, 150)
    vg = np.sin(np.radians(lon)) + 0.1 * rng.standard_normal(150)
    d, s, e = core._find_statistics(np.stack([lon, lat], 1), vg, vm.exponential_variogram_model, [0.9, 40.0, 0.1], "geographic")
    out.update(delta_geo=d, sigma_geo=s, eps_geo=e, lon=lon, lat=lat, vg=vg)
    np.savez_compressed(os.path.join(OUT, "stats_find_statistics.npz"), x=x, y=y, v=v, x3=x3, y3=y3, z3=z3, v3=v3, **out)
    print("wrote statistics fixture")


if __name__ == "__main__" and "--stats" in sys.argv:
    _import_reference(False)
    statistics()


def sklearn_callers():
    """The scikit-learn side callers of the path, run on the REAL reference: compat.Krige (compat.py:97-291) for the four
    methods, rk.RegressionKriging (rk.py:106-166), ck.ClassificationKriging (ck.py:107-192).  The learners are
    deterministic scikit-learn models, so the product (same scikit-learn) must reproduce the numbers."""
    from pykrige.ck import ClassificationKriging
    from pykrige.compat import Krige
    from pykrige.rk import RegressionKriging
    from sklearn.linear_model import LinearRegression
    from sklearn.naive_bayes import GaussianNB  # closed form: the same probabilities on every host (lbfgs-fitted learners are not)

    out = {}
    rng = np.random.default_rng(2024)
    X3 = rng.random((160, 3))
    y = np.sin(5 * X3[:, 0]) * np.cos(3 * X3[:, 1]) + 0.5 * X3[:, 2] + 0.05 * rng.standard_normal(160)
    Q3 = rng.random((50, 3))
    out.update(X3=X3, y=y, Q3=Q3)
    cases = {"ordinary": dict(variogram_model="exponential", n_closest_points=8, nlags=6),
             "universal": dict(variogram_model="linear", drift_terms=["regional_linear"]),
             "ordinary3d": dict(variogram_model="spherical", n_closest_points=12, anisotropy_scaling=(1.5, 0.7)),
             "universal3d": dict(variogram_model="gaussian", variogram_parameters=[1.0, 0.8, 0.05],
                                 drift_terms=["regional_linear"])}
    for method, kw in cases.items():
        d = 3 if method.endswith("3d") else 2
        k = Krige(method=method, **kw)
        k.fit(X3[:, :d], y)
        pts = k._dimensionality_check(Q3[:, :d], ext="points")
        pred, var = k.execute(pts)
        out["krige_%s_pred" % method], out["krige_%s_var" % method] = np.asarray(pred), np.asarray(var)
        out["krige_%s_par" % method] = np.asarray(k.model.variogram_model_parameters)
    P = np.column_stack([X3[:, 0] ** 2, X3[:, 1], np.cos(X3[:, 2])]) + 0.01 * rng.standard_normal((160, 3))
    PQ = np.column_stack([Q3[:, 0] ** 2, Q3[:, 1], np.cos(Q3[:, 2])])
    out.update(P=P, PQ=PQ)
    rk = RegressionKriging(regression_model=LinearRegression(), method="ordinary", variogram_model="spherical", n_closest_points=10)
    rk.fit(P, X3[:, :2], y)
    out["rk_pred"] = np.asarray(rk.predict(PQ, Q3[:, :2]))
    out["rk_score"] = np.array(rk.score(P[:40], X3[:40, :2] + 0.01, y[:40]))
    labels = np.digitize(y, np.quantile(y, [0.33, 0.66])).reshape(-1, 1)
    ck = ClassificationKriging(classification_model=GaussianNB(), method="ordinary",
                               variogram_model="exponential", variogram_parameters=[450.0, 0.4, 5.0], n_closest_points=10)
    # (explicit parameters: the automatic fit of the second ilr coordinate has a weakly determined nugget, and
    #  scipy.optimize.least_squares then lands on host-dependent parameters -- seen 4.3e-2 here vs 5.0e-2 on the GPU box)
    ck.fit(P, X3[:, :2], labels)
    out["labels"] = labels
    out["ck_par"] = np.array([k.model.variogram_model_parameters for k in ck.krige])
    out["ck_residual"] = np.asarray(ck.krige_residual(Q3[:, :2]))
    out["ck_pred"] = np.asarray(ck.predict(PQ, Q3[:, :2]))
    np.savez_compressed(os.path.join(OUT, "sk_callers.npz"), **out)
    print("wrote sklearn-callers fixture")


if __name__ == "__main__" and "--sk" in sys.argv:
    _import_reference(False)
    sklearn_callers()


def grid_files():
    """kriging_tools.py of the REAL reference: files it writes (kept as text fixtures) and what it reads back."""
    import pykrige.kriging_tools as kt

    d = os.path.join(OUT, "tools")
    os.makedirs(d, exist_ok=True)
    rng = np.random.default_rng(314)
    x, y = np.linspace(10.0, 55.0, 10), np.linspace(-3.0, 9.0, 7)
    xs, ys = np.linspace(10.0, 55.0, 10), np.linspace(-3.0, 27.0, 7)  # square cells for style 2
    z = rng.standard_normal((7, 10)) * 1000.0
    z[3, 3] = 123456.789
    mask = rng.random(z.shape) < 0.2
    zz = z.copy()
    zz[2, 3], zz[0, 0], zz[1, 1] = np.nan, 1.5e7, -2.5e120
    out = dict(x=x, y=y, xs=xs, ys=ys, z=z, mask=mask, zz=zz)
    kt.write_asc_grid(x, y, z, os.path.join(d, "style1.asc"), style=1)
    kt.write_asc_grid(xs, ys, np.ma.array(z, mask=mask), os.path.join(d, "style2_masked.asc"), no_data=-9999.0, style=2)
    kt.write_zmap_grid(x, y, np.ma.array(zz, mask=mask), os.path.join(d, "masked.zmap"), coord_sys="EPSG:1234")
    with open(os.path.join(d, "variant_header.asc"), "w") as f:  # lower-case keys, cell_size / nodatavalue, two footer lines
        f.write("ncols 4\nnrows 3\nxllcorner 100.0\nyllcorner 200.0\ncell_size 2.5\nnodatavalue -1\n"
                "1 2 3 4\n5 6 7 8\n9 10 11 12\nfooter line\nanother one\n")
    for tag, args in (("style1", ("style1.asc",)), ("style2", ("style2_masked.asc",)), ("variant", ("variant_header.asc", 2))):
        g, gx, gy, cell, nod = kt.read_asc_grid(os.path.join(d, args[0]), *args[1:])
        out.update({tag + "_grid": g, tag + "_x": gx, tag + "_y": gy, tag + "_cell": np.atleast_1d(np.asarray(cell, dtype=float)),
                    tag + "_nodata": nod})
    g, gx, gy, cell, nod, cs = kt.read_zmap_grid(os.path.join(d, "masked.zmap"))
    out.update(zmap_grid=g, zmap_x=gx, zmap_y=gy, zmap_cell=np.asarray(cell), zmap_nodata=nod, zmap_cs=cs)
    np.savez_compressed(os.path.join(OUT, "tools_grid_files.npz"), **out)
    print("wrote grid-file fixtures")


if __name__ == "__main__" and "--tools" in sys.argv:
    _import_reference(False)
    grid_files()


def stable_variogram(m, d):
    """The custom model of the fixture below: a 'stable' variogram psill (1 - exp(-(d / range)^1.5)) + nugget --
    not one of the six named models.  Imported by tests/ as the user-supplied callable."""
    return m[0] * (1.0 - np.exp(-((np.asarray(d) / m[1]) ** 1.5))) + m[2]


def custom_variogram():
    """variogram_model='custom' on the REAL reference (ok.py:247-254, test_core.py test_custom_variogram): OK2D grid + moving
    window + statistics, UK2D with regional_linear, OK3D."""
    from pykrige.ok import OrdinaryKriging
    from pykrige.ok3d import OrdinaryKriging3D
    from pykrige.uk import UniversalKriging

    def arr(x):
        return np.ma.getdata(x).astype(np.float64)

    par = [0.9, 0.35, 0.05]
    (x, y), v = synth(120, 180, 2)
    gx, gy = np.linspace(0, 1, 13), np.linspace(0, 1, 11)
    x[:3], y[:3] = gx[[2, 6, 10]], gy[[1, 5, 9]]
    out = dict(x=x, y=y, v=v, par=par, gx=gx, gy=gy)
    import pykrige.core as core

    ok = OrdinaryKriging(x, y, v, variogram_model="custom", variogram_parameters=par, variogram_function=stable_variogram,
                         anisotropy_scaling=1.4, anisotropy_angle=15.0)
    z, ss = ok.execute("grid", gx, gy, backend="vectorized")
    zk, ssk = ok.execute("grid", gx, gy, backend="loop", n_closest_points=9)
    # (_import_reference stubs the constructors' statistics pass; this is the real one, as ok.py:360-371 calls it)
    delta, sigma, eps = core._find_statistics(np.vstack((ok.X_ADJUSTED, ok.Y_ADJUSTED)).T, ok.Z, ok.variogram_function,
                                              ok.variogram_model_parameters, "euclidean")
    out.update(ok_z=arr(z), ok_ss=arr(ss), ok_zk=arr(zk), ok_ssk=arr(ssk), ok_eps=eps,
               ok_q=np.array([core.calcQ1(eps), core.calcQ2(eps), core.calc_cR(core.calcQ2(eps), sigma)]))
    uk = UniversalKriging(x, y, v, variogram_model="custom", variogram_parameters=par, variogram_function=stable_variogram,
                          drift_terms=["regional_linear"])
    z, ss = uk.execute("grid", gx, gy, backend="vectorized")
    out.update(uk_z=arr(z), uk_ss=arr(ss))
    (x3, y3, z3), v3 = synth(121, 150, 3)
    g3 = [np.linspace(0, 1, 6), np.linspace(0, 1, 5), np.linspace(0, 1, 4)]
    k3 = OrdinaryKriging3D(x3, y3, z3, v3, variogram_model="custom", variogram_parameters=par, variogram_function=stable_variogram)
    z, ss = k3.execute("grid", *g3, backend="vectorized")
    out.update(x3=x3, y3=y3, z3=z3, v3=v3, g3x=g3[0], g3y=g3[1], g3z=g3[2], k3_z=arr(z), k3_ss=arr(ss))
    np.savez_compressed(os.path.join(OUT, "custom_variogram.npz"), **out)
    print("wrote custom-variogram fixture")


if __name__ == "__main__" and "--custom" in sys.argv:
    _import_reference(False)
    custom_variogram()


def anisotropy():
    """core._adjust_for_anisotropy (core.py:120-193) of the REAL reference on random 2-D / 3-D inputs: the adjusted
    coordinates decide the `eps` exact-hit rule, so the product's host transform is pinned bit for bit."""
    import pykrige.core as core

    rng = np.random.default_rng(777)
    out = {}
    X2 = rng.random((500, 2)) * 100 - 30
    c2, s2, a2 = [12.5, -3.25], [2.75], [33.3]
    out.update(X2=X2, c2=c2, s2=s2, a2=a2, Y2=core._adjust_for_anisotropy(X2.copy(), c2, s2, a2))
    X3 = rng.random((500, 3)) * 10
    c3, s3, a3 = [5.0, 4.0, 3.0], [1.5, 0.4], [10.0, -25.0, 70.0]
    out.update(X3=X3, c3=c3, s3=s3, a3=a3, Y3=core._adjust_for_anisotropy(X3.copy(), c3, s3, a3))
    np.savez_compressed(os.path.join(OUT, "aniso_adjust.npz"), **out)
    print("wrote anisotropy fixture")


if __name__ == "__main__" and "--aniso" in sys.argv:
    _import_reference(False)
    anisotropy()


def geographic_spherical():
    """Round 5: coordinates_type='geographic' with the SPHERICAL model (ok.py:634-640, 990-996; variogram_models.py:56-70) -- the case
    the range-aware contraction takes since round 5 (candidates by boxes of the unit vectors).  Stations across the date line and
    up to 85 degrees of latitude, range 25 degrees: most stations are beyond the range of any grid node."""
    from pykrige.ok import OrdinaryKriging

    def arr(x):
        return np.ma.getdata(x).astype(np.float64)

    rng = np.random.default_rng(96)
    n = 500
    lon = rng.uniform(100.0, 260.0, n)  # 100 E .. 100 W through the date line
    lon = np.where(lon > 180.0, lon - 360.0, lon)
    lat = rng.uniform(-70.0, 85.0, n)
    v = np.sin(np.radians(lon) * 2) * np.cos(np.radians(lat) * 3) + 0.1 * rng.standard_normal(n)
    glon = np.concatenate([np.linspace(100.0, 180.0, 12), np.linspace(-175.0, -100.0, 11)])
    glat = np.linspace(-70.0, 85.0, 17)
    lon[:4], lat[:4] = glon[[2, 6, 13, 20]], glat[[1, 4, 8, 15]]
    params = [1.0, 25.0, 0.02]
    ok = OrdinaryKriging(lon, lat, v, variogram_model="spherical", variogram_parameters=params, coordinates_type="geographic")
    z, ss = ok.execute("grid", glon, glat, backend="vectorized")
    np.savez_compressed(os.path.join(OUT, "geo_ok2d_spherical.npz"), x=lon, y=lat, v=v, model="spherical", params_user=params, gridx=glon,
                        gridy=glat, geographic=True, z=arr(z), ss=arr(ss), A=ok._get_kriging_matrix(n))
    print("wrote geo_ok2d_spherical")


if __name__ == "__main__" and "--geo-spherical" in sys.argv:
    _import_reference(False)
    geographic_spherical()
