"""Time the moving-window path for a range of window sizes on config-2 stations (GPU box): the LDL^T solver (default)
beside the pivoting Gauss-Jordan / HBM-LU fallback (option mw_pivot = 1), whole call and solve kernels alone."""
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
print("%5s %9s | %12s %12s %14s | %12s %12s %14s | %9s" % ("k", "points", "LDLt ms", "solve ms", "points/s", "G-J/LU ms", "solve ms", "points/s", "max|dz|"))
for k, npt in ((10, 1000000), (16, 1000000), (32, 1000000), (50, 1000000), (64, 1000000), (80, 1000000), (96, 1000000), (100, 1000000), (104, 1000000), (127, 100000), (128, 100000),
               (160, 100000), (192, 100000), (200, 100000), (224, 100000), (256, 100000), (257, 20000), (320, 20000), (500, 20000), (512, 20000), (1000, 4000)):
    px, py = rng.random(npt), rng.random(npt)
    h.set_points(px, py, None)
    row, zs = [], []
    for solver in ((0,) if "--ldlt-only" in sys.argv else (0, 1)):
        h.set_option("mw_pivot", solver)
        h.predict_moving_window(k)
        t0 = time.perf_counter()
        h.predict_moving_window(k)
        dt = time.perf_counter() - t0
        row += [dt * 1e3, h.timing()["contract_ms"], npt / dt]
        zs.append(h.get_results()[0])
    if len(zs) == 1:
        print("%5d %9d | %12.2f %12.2f %14.0f |" % (k, npt, *row), flush=True)
        continue
    print("%5d %9d | %12.2f %12.2f %14.0f | %12.2f %12.2f %14.0f | %9.2e" % (k, npt, *row, np.abs(zs[0] - zs[1]).max()), flush=True)
h.set_option("mw_pivot", 0)

# many stations: only coordinates live on the device (no N x N matrix on this path)
for n, k in ((100000, 10), (1000000, 10), (1000000, 32)):
    rs = np.random.default_rng(n + k)
    sx, sy = rs.random(n), rs.random(n)
    h.set_problem(ndim=2, xs=sx, ys=sy, zs=None, values=np.sin(7 * sx) + sy, model_id=_lib.MODEL_IDS["exponential"],
                  params=internal_params("exponential", [1.0, 0.02, 0.0]))
    npt = 1000000
    h.set_points(rs.random(npt), rs.random(npt), None)
    h.predict_moving_window(k)
    t0 = time.perf_counter()
    h.predict_moving_window(k)
    dt = time.perf_counter() - t0
    print("N=%8d stations  k=%3d  npt=%8d  %9.2f ms  %10.0f points/s" % (n, k, npt, dt * 1e3, npt / dt), flush=True)
