#!/usr/bin/env python
"""Range-aware contraction: tiles of eight GATHERED 16-row groups (option "sparse_rows" 16, k_contract_spg) against aligned 128-row
blocks (128, k_contract_sp) on one box -- BASELINE config 5's per-GPU slab and a few other spherical shapes.  Prints per-phase device
times, tiles / off-diagonal K tiles / triangle products of the two forms and max |dz|, |dsigma^2| between them (and against the dense
contraction with --dense).

    python scripts/sparse_rows_ab.py [--dense] [--quick]

Also the points of every launch in Hilbert-curve order (option "sort_points") against the caller's order, on the grids and on a
shuffled point list.
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


def run(name, n, grid, params, ndim=2, seed=5, reps=3, dense=False, shuffled=0):
    """grid: axis lengths; shuffled > 0: that many random points through style='points' instead."""
    c, v = synth(seed, n, ndim)
    axes = [np.linspace(0.0, 1.0, g) for g in grid]
    if ndim == 2 and n == 8000 and grid[1] == 512:
        axes[1] = np.linspace(0.0, 1.0, 4096)[:grid[1]]  # config 5: one GPU's 512 rows of the 4096 x 4096 grid
    style, args = "grid", axes
    if shuffled:
        rng = np.random.default_rng(99)
        style, args = "points", [rng.random(shuffled) for _ in range(ndim)]
    res = {}
    # (rows of a tile, points sorted): 0 = dense contraction
    for rows, sort in (((0, 0),) if dense else ()) + ((128, 0), (16, 0), (16, 1)):
        if ndim == 2:
            m = pa.OrdinaryKriging(c[0], c[1], v, variogram_model="spherical", variogram_parameters=params)
        else:
            m = pa.OrdinaryKriging3D(c[0], c[1], c[2], v, variogram_model="spherical", variogram_parameters=params)
        h = m._get_handle()
        h.set_option("sparse", 1 if rows else 0)
        h.set_option("sort_points", sort)
        if rows:
            h.set_option("sparse_rows", rows)
        best = None
        for _ in range(1 if rows == 0 else reps):
            t0 = time.perf_counter()
            z, ss = m.execute(style, *args)
            dt = time.perf_counter() - t0
            if best is None or dt < best[0]:
                best = (dt, dict(m.last_timing))
        res[(rows, sort)] = (np.ma.getdata(z).copy(), np.ma.getdata(ss).copy(), best)
        t = best[1]
        print("%-30s rows=%3d sorted=%d  execute %8.2f ms %7.3f M pts/s | invert %6.2f rhs %6.2f contract %8.2f lists %5.2f sort %5.2f ms | "
              "tiles %d (dense %d)  ktiles %.4g  triangle products %.4g | executed %.1f TFLOP/s" % (
                  name, rows, sort, 1e3 * best[0], z.size / best[0] / 1e6, t["invert_ms"], t["rhs_ms"], t["contract_ms"],
                  t["sparse_lists_ms"], t["sort_points_ms"], t["sparse_tiles"], t["sparse_tiles_dense"], t["sparse_ktiles"],
                  t["sparse_diag_products"], t["contract_flops_executed"] / max(t["contract_ms"], 1e-9) / 1e9), flush=True)
    a, b, c2 = res[(128, 0)], res[(16, 0)], res[(16, 1)]
    print("%-30s 16 vs 128: max|dz| %.2e  max|dss| %.2e  contract %.2fx  execute %.2fx" % (
        name, np.abs(a[0] - b[0]).max(), np.abs(a[1] - b[1]).max(), a[2][1]["contract_ms"] / b[2][1]["contract_ms"], a[2][0] / b[2][0]), flush=True)
    print("%-30s sorted vs unsorted (rows 16): max|dz| %.2e  max|dss| %.2e  contract %.2fx  execute %.2fx" % (
        name, np.abs(c2[0] - b[0]).max(), np.abs(c2[1] - b[1]).max(), b[2][1]["contract_ms"] / c2[2][1]["contract_ms"], b[2][0] / c2[2][0]), flush=True)
    if dense:
        d = res[(0, 0)]
        print("%-30s sorted 16 vs dense: max|dz| %.2e  max|dss| %.2e" % (name, np.abs(d[0] - c2[0]).max(), np.abs(d[1] - c2[1]).max()), flush=True)


if __name__ == "__main__":
    os.environ.setdefault("MIK_FACTOR_CACHE", "0")
    dense = "--dense" in sys.argv
    run("config 5 slab N=8000 4096x512", 8000, (4096, 512), [1.0, 0.2, 0.01], dense=dense)
    if "--quick" not in sys.argv:
        run("N=8000 4096x512 range 0.05", 8000, (4096, 512), [1.0, 0.05, 0.01])
        run("N=8000 4096x512 range 0.6", 8000, (4096, 512), [1.0, 0.6, 0.01], reps=2)
        run("N=5000 1000x1000 range 0.3", 5000, (1000, 1000), [1.0, 0.3, 0.0], dense=dense)
        run("3-D N=2000 200x200x50 range 0.4", 2000, (200, 200, 50), [1.0, 0.4, 0.02], ndim=3)
        run("N=8000, 2e6 shuffled points", 8000, (4096, 512), [1.0, 0.2, 0.01], shuffled=2000000, reps=2)
