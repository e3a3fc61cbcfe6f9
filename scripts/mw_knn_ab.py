"""A/B of the moving-window neighbour search for small windows: lane-per-point first pass (k_mw_knn_lane, option mw_knn_lane = 1) against
the wave-per-point kernel alone (0), on config-2 stations; points as the rows of a 1000 x 1000 grid and as a shuffled list.  GPU box."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from bench import CONFIGS, synth, internal_params
from pykrige_amd import _lib

cfg = CONFIGS[2]
for n in (5000, 100000):
    rs = np.random.default_rng(cfg["seed"] if n == 5000 else 77)
    coords = [rs.random(n), rs.random(n)]
    values = np.sin(6 * coords[0]) * np.cos(4 * coords[1]) + 0.1 * rs.standard_normal(n)
    h = _lib.Handle(0)
    h.set_problem(ndim=2, xs=coords[0], ys=coords[1], zs=None, values=values, model_id=_lib.MODEL_IDS[cfg["model"]],
                  params=internal_params(cfg["model"], [1.0, 0.3 if n == 5000 else 0.05, 0.0]))
    gx = np.linspace(0.0, 1.0, 1000)
    X, Y = np.meshgrid(gx, gx)
    grid_pts = (X.ravel(), Y.ravel())
    perm = np.random.default_rng(1).permutation(X.size)
    shuf_pts = (grid_pts[0][perm], grid_pts[1][perm])
    out_pts = (grid_pts[0] * 3.0 - 1.0, grid_pts[1] * 3.0 - 1.0)  # two thirds of the points lie outside the stations' bounding box
    for label, pts in (("grid rows", grid_pts), ("shuffled", shuf_pts), ("grid 3x wider than the stations", out_pts)):
        h.set_points(pts[0], pts[1], None)
        for k in (2, 5, 10, 16, 24, 32):
            res = {}
            for lane in (0, 1):
                h.set_option("mw_knn_lane", lane)
                h.predict_moving_window(k)
                t0 = time.perf_counter()
                h.predict_moving_window(k)
                dt = time.perf_counter() - t0
                t = h.timing()
                z, ss = h.get_results()
                res[lane] = (dt * 1e3, t["rhs_ms"], t["contract_ms"], z.copy(), ss.copy())
            same = np.array_equal(res[0][3], res[1][3]) and np.array_equal(res[0][4], res[1][4])
            print("N=%6d %-32s k=%2d | wave-per-point: call %6.2f ms (search+rhs %6.2f, solve %5.2f) | lane first: call %6.2f ms (search+rhs %6.2f, solve %5.2f) | "
                  "bit-identical %s, max|dz| %.1e" % (n, label, k, res[0][0], res[0][1], res[0][2], res[1][0], res[1][1], res[1][2], same,
                                                      np.abs(res[0][3] - res[1][3]).max()), flush=True)
    h.close()
