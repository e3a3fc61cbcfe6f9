"""A/B of the moving-window LDL^T kernel's classes {G, RI} (G x G threads per point, RI x RI register tile per thread; option
mw_class): for each window size every class that covers it, solve-kernel time per 10^5 points.  GPU box."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from bench import CONFIGS, synth, internal_params
from pykrige_amd import _lib

cfg = CONFIGS[2]
coords, values = synth(cfg["seed"], cfg["n"], 2)
h = _lib.Handle(0)
h.set_problem(ndim=2, xs=coords[0], ys=coords[1], zs=None, values=values, model_id=_lib.MODEL_IDS[cfg["model"]],
              params=internal_params(cfg["model"], cfg["params"]))
rng = np.random.default_rng(0)
CLASSES = [(4, 4), (4, 6), (4, 8), (4, 10), (4, 12), (4, 13), (8, 4), (8, 6), (8, 8), (8, 10), (8, 11), (8, 12), (8, 13), (8, 14), (16, 5), (16, 6), (16, 7), (16, 8), (16, 9), (16, 10), (16, 11), (16, 12), (16, 13), (16, 14), (16, 15), (16, 16),
           (32, 5), (32, 6), (32, 7), (32, 8)]
ks = [int(a) for a in sys.argv[1:]] or [72, 80, 88, 96, 100, 104, 112, 128, 144, 160, 176, 192, 200, 208, 224, 256, 257, 320, 512, 1000]
npt = 200000
px, py = rng.random(npt), rng.random(npt)
h.set_points(px, py, None)
for k in ks:
    res = []
    if k <= 256:
        h.set_option("mw_class", 0)
        h.predict_moving_window(k)
        zref = h.get_results()[0]
    for g, ri in CLASSES + [(0, 1)]:  # (0, 1) = the blocked Cholesky kernel of the large windows ("mw_class" 1)
        if k > 256 or (g and (g * ri < k or g * ri > 1.5 * k + 16)) or (not g and k < 128):
            continue
        h.set_option("mw_class", 100 * g + ri)
        try:
            h.predict_moving_window(k)
            h.predict_moving_window(k)
        except Exception as e:
            res.append(((g, ri), None, repr(e)[:40]))
            continue
        t = h.timing()
        res.append(((g, ri), t["contract_ms"], float(np.abs(h.get_results()[0] - zref).max())))
    if k > 256:
        npt_k = 20000
        h.set_points(px[:npt_k], py[:npt_k], None)
        h.set_option("mw_class", 0)
        h.predict_moving_window(k)
        h.predict_moving_window(k)
        t = h.timing()
        print("k=%3d  kernel %d: solve %.2f ms per %d points = %.0f points/s (whole call %.2f ms)" % (k, t["mw_kernel"], t["contract_ms"], npt_k, npt_k / t["contract_ms"] * 1e3, t["predict_ms"]), flush=True)
        h.set_points(px, py, None)
        continue
    best = min((r for r in res if r[1] is not None), key=lambda r: r[1])
    print("k=%3d  best {%d,%d} %.2f ms per %d points (%.2f M points/s solve only) | " % (k, best[0][0], best[0][1], best[1], npt, npt / best[1] / 1e3)
          + "  ".join("{%d,%d} %s" % (c[0], c[1], ("%.2f" % t) if t is not None else d) for c, t, d in res)
          + " | max|dz| vs default %.1e" % max(d for _, t, d in res if t is not None), flush=True)
