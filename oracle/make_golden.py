"""Generate tests/golden/*.npz by running the REAL reference (PyKrige 1.7.3).

TEST INFRASTRUCTURE.  Run in the build container only (needs /root/reference):

    python oracle/make_golden.py            # pure-Python reference (backend='vectorized'/'loop')
    python oracle/make_golden.py --with-c   # also backend='C' (needs oracle/build_ref.sh run first)

Each fixture stores the *inputs* (stations, values, variogram, drift set-up, grid axes, mask)
and the reference's *outputs* (kriging matrix A, z, sigma^2), so tests need no reference.
Nothing from /root/reference is copied; only numbers it computed are stored.
"""
import argparse
import os
import sys

import numpy as np

REF = "/root/reference/src"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
REF_DATA = "/root/reference/tests/test_data"


def _import_reference(with_c):
    sys.path.insert(0, REF)
    import pykrige  # noqa
    import pykrige.lib
    if with_c:
        pykrige.lib.__path__.append(os.path.join(HERE, "_ref", "pykrige_lib"))
        import pykrige.lib.cok  # noqa  (fails loudly if build_ref.sh has not been run)
    import pykrige.ok, pykrige.uk, pykrige.ok3d, pykrige.uk3d
    stub = lambda *a, **k: (np.zeros(2), np.ones(2), np.zeros(2))  # O(N^4) ctor statistics
    for m in (pykrige.ok, pykrige.uk, pykrige.ok3d, pykrige.uk3d):
        m._find_statistics = stub
    return pykrige


def synth(seed, n, ndim):
    rng = np.random.default_rng(seed)
    c = [rng.random(n) for _ in range(ndim)]
    v = np.sin(6 * c[0]) * np.cos(4 * c[1])
    if ndim == 3:
        v = v * np.cos(3 * c[2])
    v = v + 0.1 * rng.standard_normal(n)
    return c, v


def put_on_nodes(coords, axes, k, seed):
    """Overwrite k stations with grid-node coordinates (exercises the eps rule)."""
    rng = np.random.default_rng(seed + 1000)
    shape = tuple(len(ax) for ax in axes)
    flat = rng.choice(int(np.prod(shape)), size=k, replace=False)  # distinct nodes (else A is singular)
    idx = np.unravel_index(flat, shape)
    for c, ax, ii in zip(coords, axes, idx):
        c[:k] = np.asarray(ax)[ii]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--with-c", action="store_true")
    args = ap.parse_args()
    pk = _import_reference(args.with_c)
    from pykrige.ok import OrdinaryKriging
    from pykrige.uk import UniversalKriging
    from pykrige.ok3d import OrdinaryKriging3D
    from pykrige.uk3d import UniversalKriging3D
    os.makedirs(OUT, exist_ok=True)
    made = []

    def save(name, **kw):
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **kw)
        made.append(name)

    def arr(x):
        return np.ma.getdata(x).astype(np.float64)

    # ---- reference test data (tests/test_core.py:490-507 etc.) ---------------------------
    data = np.genfromtxt(os.path.join(REF_DATA, "test_data.txt"))
    # test_ok: exponential [500, 3000, 0], grid from the .asc header (test_core.py:28-42, 490-507)
    import pykrige.kriging_tools as kt
    ans, gridx, gridy, cellsize, no_data = kt.read_asc_grid(os.path.join(REF_DATA, "test1_answer.asc"), footer=2)
    ok = OrdinaryKriging(data[:, 0], data[:, 1], data[:, 2], variogram_model="exponential",
                         variogram_parameters=[500.0, 3000.0, 0.0])
    z, ss = ok.execute("grid", gridx, gridy, backend="vectorized")
    save("ref_test_ok", x=data[:, 0], y=data[:, 1], v=data[:, 2], model="exponential",
         params_user=[500.0, 3000.0, 0.0], gridx=gridx, gridy=gridy, z=arr(z), ss=arr(ss),
         answer=ans, A=ok._get_kriging_matrix(data.shape[0]))
    # test_uk: regional_linear, same variogram (test_core.py:707-725)
    ans2, gridx2, gridy2, _, _ = kt.read_asc_grid(os.path.join(REF_DATA, "test2_answer.asc"), footer=2)
    uk = UniversalKriging(data[:, 0], data[:, 1], data[:, 2], variogram_model="exponential",
                          variogram_parameters=[500.0, 3000.0, 0.0], drift_terms=["regional_linear"])
    z, ss = uk.execute("grid", gridx2, gridy2, backend="vectorized")
    save("ref_test_uk", x=data[:, 0], y=data[:, 1], v=data[:, 2], model="exponential",
         params_user=[500.0, 3000.0, 0.0], gridx=gridx2, gridy=gridy2, z=arr(z), ss=arr(ss),
         answer=ans2, regional_linear=True)
    # test_ok3d: KT3D answer incl. sigma^2 (test_core.py:1914-1989)
    d3 = np.genfromtxt(os.path.join(REF_DATA, "test3d_data.txt"), skip_header=1)
    a3 = np.genfromtxt(os.path.join(REF_DATA, "test3d_answer.txt"))
    k3d = OrdinaryKriging3D(d3[:, 0], d3[:, 1], d3[:, 2], d3[:, 3], variogram_model="linear",
                            variogram_parameters=[1.0, 0.1])
    g3 = np.arange(10.0)
    z, ss = k3d.execute("grid", g3, g3, g3, backend="vectorized")
    save("ref_test_ok3d", x=d3[:, 0], y=d3[:, 1], zc=d3[:, 2], v=d3[:, 3], model="linear",
         params_user=[1.0, 0.1], gridx=g3, gridy=g3, gridz=g3, z=arr(z), ss=arr(ss),
         answer_z=a3[:, 0].reshape((10, 10, 10)), answer_ss=a3[:, 1].reshape((10, 10, 10)))

    # ---- synthetic families --------------------------------------------------
    MODELS = {  # user list-form parameters
        "linear": [1.5, 0.05], "power": [1.2, 1.4, 0.05], "gaussian": [1.0, 0.4, 0.02],
        "spherical": [1.0, 0.5, 0.05], "exponential": [1.0, 0.3, 0.0], "hole-effect": [1.0, 0.6, 0.05],
    }
    # OK2D, every variogram model, exact-hit nodes, anisotropy on one of them, with and without exact_values
    for i, (model, par) in enumerate(MODELS.items()):
        n = 120
        (x, y), v = synth(10 + i, n, 2)
        gx_, gy_ = np.linspace(0, 1, 23), np.linspace(0, 1, 17)
        put_on_nodes([x, y], [gx_, gy_], 6, 10 + i)
        aniso = dict(anisotropy_scaling=3.0, anisotropy_angle=45.0) if model in ("spherical", "exponential") else {}
        for exact in (True, False):
            ok = OrdinaryKriging(x, y, v, variogram_model=model, variogram_parameters=list(par),
                                 exact_values=exact, **aniso)
            z, ss = ok.execute("grid", gx_, gy_, backend="vectorized")
            zl, ssl = ok.execute("grid", gx_, gy_, backend="loop")
            extra = {}
            if args.with_c and model != "hole-effect":
                zc, ssc = ok.execute("grid", gx_, gy_, backend="C")
                extra = dict(z_c=arr(zc), ss_c=arr(ssc))
            save("ok2d_%s_%s" % (model.replace("-", ""), "exact" if exact else "noexact"),
                 x=x, y=y, v=v, model=model, params_user=par, exact=exact, gridx=gx_, gridy=gy_,
                 scaling=aniso.get("anisotropy_scaling", 1.0), angle=aniso.get("anisotropy_angle", 0.0),
                 z=arr(z), ss=arr(ss), z_loop=arr(zl), ss_loop=arr(ssl),
                 A=ok._get_kriging_matrix(n), xadj=ok.X_ADJUSTED, yadj=ok.Y_ADJUSTED, **extra)
    # OK2D masked + points styles, N=500
    (x, y), v = synth(30, 500, 2)
    gx_, gy_ = np.linspace(0, 1, 31), np.linspace(0, 1, 29)
    put_on_nodes([x, y], [gx_, gy_], 8, 30)
    ok = OrdinaryKriging(x, y, v, variogram_model="exponential", variogram_parameters=[1.0, 0.3, 0.0])
    rng = np.random.default_rng(31)
    mask = rng.random((29, 31)) < 0.3
    zm, ssm = ok.execute("masked", gx_, gy_, mask=mask, backend="vectorized")
    px, py = rng.random(200), rng.random(200)
    px[:4], py[:4] = x[:4], y[:4]
    zp, ssp = ok.execute("points", px, py, backend="vectorized")
    save("ok2d_masked_points", x=x, y=y, v=v, model="exponential", params_user=[1.0, 0.3, 0.0],
         gridx=gx_, gridy=gy_, mask=mask, z_masked=arr(zm), ss_masked=arr(ssm),
         px=px, py=py, z_points=arr(zp), ss_points=arr(ssp))
    # OK2D N=2000 (large-N behaviour), 48x40 grid
    (x, y), v = synth(2, 2000, 2)
    gx_, gy_ = np.linspace(0, 1, 48), np.linspace(0, 1, 40)
    put_on_nodes([x, y], [gx_, gy_], 8, 2)
    ok = OrdinaryKriging(x, y, v, variogram_model="exponential", variogram_parameters=[1.0, 0.3, 0.0])
    z, ss = ok.execute("grid", gx_, gy_, backend="vectorized")
    save("ok2d_n2000", x=x, y=y, v=v, model="exponential", params_user=[1.0, 0.3, 0.0],
         gridx=gx_, gridy=gy_, z=arr(z), ss=arr(ss))
    # UK2D regional_linear + point_log (config-4 shape), one well on a grid node; + specified + functional
    wells = [[0.3137, 0.7219, 1.0], [0.6621, 0.2483, -0.5], [0.8412, 0.8127, 2.0]]
    for name, n, wl in (("uk2d_rl_pl", 400, wells), ("uk2d_rl_pl_node", 150, wells[:2] + [[0.5, 0.25, 2.0]])):
        (x, y), v = synth(4 if n == 400 else 41, n, 2)
        gx_, gy_ = np.linspace(0, 1, 33), np.linspace(0, 1, 21)
        put_on_nodes([x, y], [gx_, gy_], 8, 4)
        uk = UniversalKriging(x, y, v, variogram_model="exponential", variogram_parameters=[1.0, 0.3, 0.01],
                              drift_terms=["regional_linear", "point_log"], point_drift=wl,
                              anisotropy_scaling=(2.0 if n == 150 else 1.0), anisotropy_angle=(30.0 if n == 150 else 0.0))
        z, ss = uk.execute("grid", gx_, gy_, backend="vectorized")
        save(name, x=x, y=y, v=v, model="exponential", params_user=[1.0, 0.3, 0.01], gridx=gx_, gridy=gy_,
             wells=np.array(wl), regional_linear=True, scaling=uk.anisotropy_scaling, angle=uk.anisotropy_angle,
             z=arr(z), ss=arr(ss), A=uk._get_kriging_matrix(n, n + 2 + len(wl)))
    (x, y), v = synth(42, 200, 2)
    gx_, gy_ = np.linspace(0, 1, 19), np.linspace(0, 1, 15)
    spec_data = np.sin(3 * x) + y
    GX, GY = np.meshgrid(gx_, gy_)
    spec_grid = np.sin(3 * GX) + GY
    funcs = [lambda a, b: a * b, lambda a, b: a**2]
    uk = UniversalKriging(x, y, v, variogram_model="spherical", variogram_parameters=[1.0, 0.5, 0.05],
                          drift_terms=["regional_linear", "specified", "functional"], specified_drift=[spec_data],
                          functional_drift=funcs)
    z, ss = uk.execute("grid", gx_, gy_, backend="vectorized", specified_drift_arrays=[spec_grid])
    save("uk2d_spec_func", x=x, y=y, v=v, model="spherical", params_user=[1.0, 0.5, 0.05], gridx=gx_, gridy=gy_,
         spec_data=spec_data, spec_grid=spec_grid, z=arr(z), ss=arr(ss))
    # OK3D gaussian with nugget, all three angles (config-3 shape)
    (x, y, zc), v = synth(3, 300, 3)
    gx_, gy_, gz_ = np.linspace(0, 1, 11), np.linspace(0, 1, 9), np.linspace(0, 1, 7)
    put_on_nodes([x, y, zc], [gx_, gy_, gz_], 8, 3)
    k3 = OrdinaryKriging3D(x, y, zc, v, variogram_model="gaussian", variogram_parameters=[1.0, 0.4, 0.02],
                           anisotropy_scaling_y=1.5, anisotropy_scaling_z=2.0, anisotropy_angle_x=10.0,
                           anisotropy_angle_y=20.0, anisotropy_angle_z=30.0)
    z, ss = k3.execute("grid", gx_, gy_, gz_, backend="vectorized")
    rng = np.random.default_rng(33)
    mask3 = rng.random((7, 9, 11)) < 0.25
    zm, ssm = k3.execute("masked", gx_, gy_, gz_, mask=mask3, backend="vectorized")
    save("ok3d_gaussian_aniso", x=x, y=y, zc=zc, v=v, model="gaussian", params_user=[1.0, 0.4, 0.02],
         gridx=gx_, gridy=gy_, gridz=gz_, scaling=[1.5, 2.0], angle=[10.0, 20.0, 30.0], z=arr(z), ss=arr(ss),
         mask=mask3, z_masked=arr(zm), ss_masked=arr(ssm), A=k3._get_kriging_matrix(300))
    # UK3D regional_linear + functional
    (x, y, zc), v = synth(5, 250, 3)
    u3 = UniversalKriging3D(x, y, zc, v, variogram_model="exponential", variogram_parameters=[1.0, 0.5, 0.02],
                            drift_terms=["regional_linear", "functional"], functional_drift=[lambda a, b, c: a * c])
    z, ss = u3.execute("grid", gx_, gy_, gz_, backend="vectorized")
    save("uk3d_rl_func", x=x, y=y, zc=zc, v=v, model="exponential", params_user=[1.0, 0.5, 0.02],
         gridx=gx_, gridy=gy_, gridz=gz_, z=arr(z), ss=arr(ss))
    print("wrote", len(made), "fixtures to", OUT)
    tot = sum(os.path.getsize(os.path.join(OUT, m + ".npz")) for m in made)
    print("total bytes", tot)


if __name__ == "__main__":
    main()
