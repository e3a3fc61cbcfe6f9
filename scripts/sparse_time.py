#!/usr/bin/env python
"""Range-aware contraction (option "sparse") against the dense contraction on one box: BASELINE config 5's per-GPU slab
(OK2D N=8000, 4096 x 512, spherical [1, 0.2, 0.01]) and a few other spherical shapes.  Prints per-phase device times, the tiles
contracted / tiles of the dense form, and max |dz|, |dsigma^2| between the two paths.

    python scripts/sparse_time.py [--quick]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pykrige_amd as pa  # noqa: E402


def synth(seed, n, ndim):
    rng = np.random.default_rng(seed)
    c = [rng.random(n) for _ in range(ndim)]
    v = np.sin(6 * c[0]) * np.cos(4 * c[1])
    if ndim == 3:
        v = v * np.cos(3 * c[2])
    return c, v + 0.1 * rng.standard_normal(n)


def run(name, n, grid, params, ndim=2, seed=5, reps=3, modes=(0, 1, 11)):
    c, v = synth(seed, n, ndim)
    axes = [np.linspace(0.0, 1.0, g) for g in grid]
    if ndim == 2 and n == 8000 and grid[1] == 512:
        axes[1] = np.linspace(0.0, 1.0, 4096)[:grid[1]]  # config 5: one GPU's 512 rows of the 4096 x 4096 grid
    res = {}
    for sparse in modes:
        if ndim == 2:
            m = pa.OrdinaryKriging(c[0], c[1], v, variogram_model="spherical", variogram_parameters=params)
        else:
            m = pa.OrdinaryKriging3D(c[0], c[1], c[2], v, variogram_model="spherical", variogram_parameters=params)
        m._get_handle().set_option("sparse", 1 if sparse else 0)  # mode 11 = sparse with ONE lane (option sparse_lanes), 1 = the default two
        m._get_handle().set_option("sparse_lanes", 1 if sparse == 11 else 2)
        best = None
        for _ in range(reps):
            t0 = time.perf_counter()
            z, ss = m.execute("grid", *axes)
            dt = time.perf_counter() - t0
            if best is None or dt < best[0]:
                best = (dt, dict(m.last_timing))
        res[sparse] = (np.ma.getdata(z).copy(), np.ma.getdata(ss).copy(), best)
        t = best[1]
        npt = z.size
        print("%-34s sparse=%2d  execute %8.2f ms  %7.3f M points/s | invert %6.2f rhs %7.2f contract %8.2f lists %5.2f ms | "
              "tiles %d / %d  ktiles %.3g / %.3g | executed %.1f TFLOP/s" % (
                  name, sparse, 1e3 * best[0], npt / best[0] / 1e6, t["invert_ms"], t["rhs_ms"], t["contract_ms"], t["sparse_lists_ms"],
                  t["sparse_tiles"], t["sparse_tiles_dense"], t["sparse_ktiles"], t["sparse_ktiles_dense"],
                  t["contract_flops_executed"] / max(t["contract_ms"], 1e-9) / 1e9), flush=True)
    if 0 in res and 1 in res:
        print("%-34s max|dz| %.2e  max|dss| %.2e  speed-up %.2fx" % (
            name, np.abs(res[0][0] - res[1][0]).max(), np.abs(res[0][1] - res[1][1]).max(), res[0][2][0] / res[1][2][0]), flush=True)


if __name__ == "__main__":
    quick = "--quick" in sys.argv
    run("c5 slab N=8000 4096x512 r=0.2", 8000, (4096, 512), [1.0, 0.2, 0.01])
    run("c5 bench grid (y over 0..1)", 8000, (4096, 513), [1.0, 0.2, 0.01])
    if not quick:
        run("N=8000 4096x512 r=0.05", 8000, (4096, 512), [1.0, 0.05, 0.01])
        run("N=8000 4096x512 r=0.6", 8000, (4096, 512), [1.0, 0.6, 0.01])
        run("N=8000 4096x512 r=2 (all active)", 8000, (4096, 512), [1.0, 2.0, 0.01])
        run("N=5000 1000x1000 r=0.3", 5000, (1000, 1000), [1.0, 0.3, 0.0], seed=2)
        run("N=2000 3-D 200x200x50 r=0.4", 2000, (200, 200, 50), [1.0, 0.4, 0.02], ndim=3, seed=3)
        run("c1 N=100 50x50 r=0.5", 100, (50, 50), [1.0, 0.5, 0.05], seed=1)
