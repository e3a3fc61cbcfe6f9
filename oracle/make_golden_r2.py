"""Round-2 golden vectors from the REAL reference (PyKrige 1.7.3 at /root/reference/src) -> tests/golden/r2_host_rules.npz:

  zs_*   external_Z bilinear look-up on ascending / descending / unsorted axes (uk.py:512-628: the index rule
         min{i: g_i >= v}, max{i: g_i <= v} is applied to the axes AS GIVEN)
  upd_*  update_variogram_model call sequences (ok.py:379-545, uk.py:630-760, ok3d.py:368-520): omitted anisotropy
         keywords reset the object to isotropic; UniversalKriging keeps the wells' construction-time adjustment
  pst_*  _find_statistics(..., pseudo_inv=True) on stations with duplicates (core.py:749-752, 759-836)

TEST INFRASTRUCTURE; run in the build container only:  python oracle/make_golden_r2.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import OUT, _import_reference, synth  # noqa: E402


def main():
    _import_reference(False)
    from pykrige import core as rcore
    from pykrige import variogram_models as rvm
    from pykrige.ok import OrdinaryKriging
    from pykrige.ok3d import OrdinaryKriging3D
    from pykrige.uk import UniversalKriging

    def arr(x):
        return np.ma.getdata(x).astype(np.float64)

    out = {}
    # ---- z-scalars ------------------------------------------------------------------------------------
    (x, y), v = synth(91, 40, 2)
    ex, ey = np.linspace(-0.1, 1.1, 13), np.linspace(-0.2, 1.2, 17)
    EX, EY = np.meshgrid(ex, ey)
    dem = np.sin(2 * EX) + EY**2
    rng = np.random.default_rng(92)
    perm_x, perm_y = rng.permutation(ex.size), rng.permutation(ey.size)
    qx, qy = rng.random(60), rng.random(60)
    qx[:3], qy[:3] = ex[[2, 5, 7]], ey[[3, 3, 9]]  # on nodes / on a grid line
    qy[3] = ey[6]
    variants = {"asc": (ex, ey, dem), "descy": (ex, ey[::-1], dem[::-1, :]), "descxy": (ex[::-1], ey[::-1], dem[::-1, ::-1]),
                "perm": (ex[perm_x], ey[perm_y], dem[np.ix_(perm_y, perm_x)])}
    out["zs_qx"], out["zs_qy"], out["zs_x"], out["zs_y"], out["zs_v"] = qx, qy, x, y, v
    for name, (ax, ay, dd) in variants.items():
        uk = UniversalKriging(x, y, v, variogram_model="exponential", variogram_parameters=[1.0, 0.3, 0.02],
                              drift_terms=["external_Z"], external_drift=dd, external_drift_x=ax, external_drift_y=ay)
        out["zs_%s_ax" % name], out["zs_%s_ay" % name], out["zs_%s_dem" % name] = ax, ay, dd
        out["zs_%s_points" % name] = uk._calculate_data_point_zscalars(qx, qy)
        out["zs_%s_stations" % name] = uk.z_scalars
        z, ss = uk.execute("points", qx, qy, backend="vectorized")
        out["zs_%s_z" % name], out["zs_%s_ss" % name] = arr(z), arr(ss)
    # ---- update_variogram_model sequences ---------------------------------------------------------------
    (x, y), v = synth(93, 90, 2)
    gx_, gy_ = np.linspace(0, 1, 9), np.linspace(0, 1, 7)
    out["upd_x"], out["upd_y"], out["upd_v"], out["upd_gx"], out["upd_gy"] = x, y, v, gx_, gy_
    ok = OrdinaryKriging(x, y, v, variogram_model="exponential", variogram_parameters=[1.0, 0.3, 0.0],
                         anisotropy_scaling=3.0, anisotropy_angle=45.0)
    ok.update_variogram_model("spherical", [1.0, 0.5, 0.05])  # anisotropy omitted -> back to isotropic
    z, ss = ok.execute("grid", gx_, gy_, backend="vectorized")
    out["upd_ok_reset_z"], out["upd_ok_reset_ss"] = arr(z), arr(ss)
    ok.update_variogram_model("spherical", [1.0, 0.5, 0.05], anisotropy_scaling=2.0, anisotropy_angle=20.0)
    z, ss = ok.execute("grid", gx_, gy_, backend="vectorized")
    out["upd_ok_set_z"], out["upd_ok_set_ss"] = arr(z), arr(ss)
    wells = [[0.31, 0.72, 1.0], [0.66, 0.25, -0.5]]
    uk = UniversalKriging(x, y, v, variogram_model="exponential", variogram_parameters=[1.0, 0.3, 0.01],
                          drift_terms=["regional_linear", "point_log"], point_drift=wells, anisotropy_scaling=2.0,
                          anisotropy_angle=30.0)
    uk.update_variogram_model("exponential", [1.0, 0.3, 0.01])  # stations re-adjusted to isotropic, wells are NOT
    z, ss = uk.execute("grid", gx_, gy_, backend="vectorized")
    out["upd_wells"] = np.array(wells)
    out["upd_uk_reset_z"], out["upd_uk_reset_ss"] = arr(z), arr(ss)
    (x3, y3, z3), v3 = synth(94, 70, 3)
    gz_ = np.linspace(0, 1, 4)
    out["upd_x3"], out["upd_y3"], out["upd_z3"], out["upd_v3"], out["upd_gz"] = x3, y3, z3, v3, gz_
    k3 = OrdinaryKriging3D(x3, y3, z3, v3, variogram_model="gaussian", variogram_parameters=[1.0, 0.4, 0.02],
                           anisotropy_scaling_y=1.5, anisotropy_scaling_z=2.0, anisotropy_angle_x=10.0,
                           anisotropy_angle_y=20.0, anisotropy_angle_z=30.0)
    k3.update_variogram_model("gaussian", [1.0, 0.4, 0.02], anisotropy_scaling_z=2.0)
    z, ss = k3.execute("grid", gx_, gy_, gz_, backend="vectorized")
    out["upd_ok3d_z"], out["upd_ok3d_ss"] = arr(z), arr(ss)
    # ---- statistics with pseudo_inv on duplicated stations -------------------------------------------------
    (x, y), v = synth(95, 48, 2)
    x[10], y[10] = x[3], y[3]
    x[25], y[25] = x[7], y[7]
    x[40], y[40] = x[3], y[3]
    X = np.stack([x, y], 1)
    out["pst_x"], out["pst_y"], out["pst_v"] = x, y, v
    for model, fn, par in (("linear", rvm.linear_variogram_model, [1.5, 0.0]),
                           ("exponential", rvm.exponential_variogram_model, [1.0, 0.3, 0.0]),
                           ("spherical", rvm.spherical_variogram_model, [0.95, 0.5, 0.05])):
        d, s, e = rcore._find_statistics(X, v, fn, par, "euclidean", True)
        out["pst_%s_par" % model] = np.array(par)
        out["pst_%s_delta" % model], out["pst_%s_sigma" % model], out["pst_%s_eps" % model] = d, s, e
    import pykrige.ok

    pykrige.ok._find_statistics = rcore._find_statistics  # make_golden stubs it in the class modules; the real one here
    okp = OrdinaryKriging(x, y, v, variogram_model="linear", variogram_parameters=[1.5, 0.0], pseudo_inv=True,
                          enable_statistics=True)
    out["pst_class_Q1"], out["pst_class_Q2"], out["pst_class_cR"] = okp.Q1, okp.Q2, okp.cR
    np.savez_compressed(os.path.join(OUT, "r2_host_rules.npz"), **out)
    print("wrote r2_host_rules.npz with", len(out), "arrays,", os.path.getsize(os.path.join(OUT, "r2_host_rules.npz")), "bytes")


if __name__ == "__main__":
    main()
