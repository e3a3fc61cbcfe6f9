"""Full-size parity fixtures: BASELINE configs 2-5 at their real station counts, kriged by the REAL reference.

TEST INFRASTRUCTURE.  Run in the build container only (needs /root/reference; --with-c needs oracle/build_ref.sh):

    python oracle/make_golden_fullsize.py --with-c

For each config (seeds, station counts, variograms, drift set-up) one contiguous row slab of the
config's own grid with >= 16 384 points is kriged by PyKrige 1.7.3's `execute("grid", ..., backend="vectorized")`
(config 2 also `backend="C"`); 8 stations are first moved onto grid nodes of that slab so the eps rule
(ok.py:665-672) is exercised at full size.  Stored: the stations as used, the slab's axes, the reference's z and
sigma^2, and the 1-norm condition number of the kriging matrix (the tolerance's context).  Nothing from
/root/reference is copied; only numbers it computed are stored.  tests/test_hip_parity.py compares the HIP path
with these at 1e-8 / 1e-6.
"""
import argparse
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import _import_reference, synth  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "fullsize")

WELLS = [[0.3137, 0.7219, 1.0], [0.6621, 0.2483, -0.5], [0.8412, 0.8127, 2.0]]
# name: seed, n, grid sizes (x, y[, z]), slab = index ranges into the y (2-D) or y and z (3-D) axes
CASES = {
    "c2": dict(ndim=2, seed=2, n=5000, grid=(1000, 1000), model="exponential", params=[1.0, 0.3, 0.0], rows=(500, 517)),
    "c3": dict(ndim=3, seed=3, n=2000, grid=(200, 200, 50), model="gaussian", params=[1.0, 0.4, 0.02], rows=(60, 142),
               zrows=(25, 26)),
    "c4": dict(ndim=2, seed=4, n=4000, grid=(1024, 1024), model="exponential", params=[1.0, 0.3, 0.01], rows=(300, 316),
               rl=True, wells=WELLS),
    "c5": dict(ndim=2, seed=5, n=8000, grid=(4096, 4096), model="spherical", params=[1.0, 0.2, 0.01], rows=(2048, 2052)),
    # round 3: the fourth class at full size -- UniversalKriging3D (uk3d.py:739-811), regional_linear + one functional drift
    # (f(x, y, z) = x z on ADJUSTED coordinates, tests/_fixtures.py FUNCS["uk3d"]), anisotropic, N = 2000
    "uk3d": dict(ndim=3, seed=7, n=2000, grid=(200, 200, 50), model="exponential", params=[1.0, 0.5, 0.02], rows=(40, 122),
                 zrows=(30, 31), uk3d=True, scaling=(1.5, 0.7), angle=(10.0, 20.0, 30.0)),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--with-c", action="store_true")
    ap.add_argument("--only", default=None)
    args = ap.parse_args()
    _import_reference(args.with_c)
    from pykrige.ok import OrdinaryKriging
    from pykrige.ok3d import OrdinaryKriging3D
    from pykrige.uk import UniversalKriging
    from pykrige.uk3d import UniversalKriging3D

    os.makedirs(OUT, exist_ok=True)
    for name, c in CASES.items():
        if args.only and name not in args.only.split(","):
            continue
        t0 = time.time()
        ndim = c["ndim"]
        coords, v = synth(c["seed"], c["n"], ndim)
        axes = [np.linspace(0.0, 1.0, g) for g in c["grid"]]
        slab = [axes[0], axes[1][c["rows"][0]:c["rows"][1]]]
        if ndim == 3:
            slab.append(axes[2][c["zrows"][0]:c["zrows"][1]])
        # 8 stations onto distinct nodes of the slab
        rng = np.random.default_rng(c["seed"] + 1000)
        shape = tuple(len(a) for a in slab)
        flat = rng.choice(int(np.prod(shape)), size=8, replace=False)
        idx = np.unravel_index(flat, shape)
        for cc, ax, ii in zip(coords, slab, idx):
            cc[:8] = ax[ii]
        extra = {}
        if c.get("uk3d"):
            sc, an = c["scaling"], c["angle"]
            k = UniversalKriging3D(coords[0], coords[1], coords[2], v, variogram_model=c["model"],
                                   variogram_parameters=list(c["params"]), drift_terms=["regional_linear", "functional"],
                                   functional_drift=[lambda a, b, cc: a * cc], anisotropy_scaling_y=sc[0], anisotropy_scaling_z=sc[1],
                                   anisotropy_angle_x=an[0], anisotropy_angle_y=an[1], anisotropy_angle_z=an[2])
            A = k._get_kriging_matrix(c["n"], c["n"] + 4)
            z, ss = k.execute("grid", slab[0], slab[1], slab[2], backend="vectorized")
            extra = dict(regional_linear=True, scaling=np.array(sc), angle=np.array(an))
        elif ndim == 3:
            k = OrdinaryKriging3D(coords[0], coords[1], coords[2], v, variogram_model=c["model"],
                                  variogram_parameters=list(c["params"]))
            A = k._get_kriging_matrix(c["n"])
            z, ss = k.execute("grid", slab[0], slab[1], slab[2], backend="vectorized")
        elif c.get("rl"):
            k = UniversalKriging(coords[0], coords[1], v, variogram_model=c["model"], variogram_parameters=list(c["params"]),
                                 drift_terms=["regional_linear", "point_log"], point_drift=c["wells"])
            A = k._get_kriging_matrix(c["n"], c["n"] + 2 + len(c["wells"]))
            z, ss = k.execute("grid", slab[0], slab[1], backend="vectorized")
            extra = dict(wells=np.array(c["wells"]), regional_linear=True)
        else:
            k = OrdinaryKriging(coords[0], coords[1], v, variogram_model=c["model"], variogram_parameters=list(c["params"]))
            A = k._get_kriging_matrix(c["n"])
            z, ss = k.execute("grid", slab[0], slab[1], backend="vectorized")
            if args.with_c and name == "c2":
                zc, ssc = k.execute("grid", slab[0], slab[1], backend="C")
                extra = dict(z_c=np.asarray(zc, dtype=np.float64), ss_c=np.asarray(ssc, dtype=np.float64))
        cond1 = float(np.linalg.cond(A, 1))
        z, ss = np.ma.getdata(z).astype(np.float64), np.ma.getdata(ss).astype(np.float64)
        kw = dict(x=coords[0], y=coords[1], v=v, model=c["model"], params_user=np.array(c["params"]), gridx=slab[0],
                  gridy=slab[1], z=z, ss=ss, cond1=cond1, node_flat=flat, grid=np.array(c["grid"]), rows=np.array(c["rows"]),
                  **extra)
        if ndim == 3:
            kw.update(zc=coords[2], gridz=slab[2], zrows=np.array(c["zrows"]))
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **kw)
        print("%s: %d points, cond1(A) = %.3g, %d bytes, %.0f s" % (
            name, z.size, cond1, os.path.getsize(os.path.join(OUT, name + ".npz")), time.time() - t0), flush=True)


def moving_window_c2():
    """round 4: the moving window at BASELINE config 2's station count -- the stations and the row slab of fullsize/c2.npz (17 rows
    x 1000, 8 stations on nodes), kriged by the reference's backend='C' (cKDTree.query + lib/cok.pyx _c_exec_loop_moving_window,
    ok.py:929-986) with n_closest_points = 10 and 100; bench.py checks its moving-window lines against these."""
    _import_reference(True)
    from pykrige.ok import OrdinaryKriging

    g = np.load(os.path.join(OUT, "c2.npz"))
    k = OrdinaryKriging(g["x"], g["y"], g["v"], variogram_model=str(g["model"]), variogram_parameters=g["params_user"].tolist())
    kw = dict(x=g["x"], y=g["y"], v=g["v"], model=str(g["model"]), params_user=g["params_user"], gridx=g["gridx"], gridy=g["gridy"],
              windows=np.array([10, 100]))
    for w in (10, 100):
        t0 = time.time()
        z, ss = k.execute("grid", g["gridx"], g["gridy"], backend="C", n_closest_points=w)
        kw["z_k%d" % w], kw["ss_k%d" % w] = np.asarray(z, dtype=np.float64), np.asarray(ss, dtype=np.float64)
        print("mw_c2 k=%d: %d points, %.0f s" % (w, z.size, time.time() - t0), flush=True)
    np.savez_compressed(os.path.join(OUT, "mw_c2.npz"), **kw)


if __name__ == "__main__":
    if "--moving-window" in sys.argv:
        moving_window_c2()
    else:
        main()
