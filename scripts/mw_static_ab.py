"""A/B of the moving-window LDL^T kernel with the variogram model as a compile-time constant (option mw_static = 1, round 4) against the
dynamic form (0): whole call and solve kernel per 10^6 (k <= 104) / 10^5 points on config-2 stations, random points.  GPU box."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from bench import CONFIGS, synth, internal_params
from pykrige_amd import _lib

cfg = CONFIGS[2]
coords, values = synth(cfg["seed"], cfg["n"], 2)
rng = np.random.default_rng(0)
for model, params in (("exponential", [1.0, 0.3, 0.0]), ("spherical", [1.0, 0.3, 0.01]), ("gaussian", [1.0, 0.3, 0.02])):
    h = _lib.Handle(0)
    h.set_problem(ndim=2, xs=coords[0], ys=coords[1], zs=None, values=values, model_id=_lib.MODEL_IDS[model], params=internal_params(model, params))
    for k, npt in ((10, 1000000), (16, 1000000), (24, 1000000), (32, 1000000), (50, 1000000), (64, 1000000), (100, 1000000), (128, 100000), (200, 100000), (256, 100000)):
        if model != "exponential" and k not in (10, 50, 100):
            continue
        px, py = rng.random(npt), rng.random(npt)
        h.set_points(px, py, None)
        res = {}
        for static in (0, 1):
            h.set_option("mw_static", static)
            h.predict_moving_window(k)
            t0 = time.perf_counter()
            h.predict_moving_window(k)
            dt = time.perf_counter() - t0
            res[static] = (dt * 1e3, h.timing()["contract_ms"], h.get_results()[0].copy(), h.get_results()[1].copy())
        print("%-11s k=%3d %8d points | dynamic: call %7.2f ms solve %7.2f | model compiled in: call %7.2f ms solve %7.2f (%.2fx) | max|dz| %.1e max|dss| %.1e" % (
            model, k, npt, res[0][0], res[0][1], res[1][0], res[1][1], res[0][1] / res[1][1], np.abs(res[0][2] - res[1][2]).max(),
            np.abs(res[0][3] - res[1][3]).max()), flush=True)
    h.close()
