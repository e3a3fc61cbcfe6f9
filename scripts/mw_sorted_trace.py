"""One moving-window call (k = 10, 10^6 points) over a shuffled point list sorted on the device and over the rows of a grid, for
`rocprofv3 --kernel-trace --stats` (scripts/gpu_r04b_14.sh)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import CONFIGS, synth, internal_params
from pykrige_amd import _lib
cfg = CONFIGS[2]
coords, values = synth(cfg["seed"], cfg["n"], 2)
rng = np.random.default_rng(0)
npt = 1000000
px, py = rng.random(npt), rng.random(npt)
gx = np.linspace(0, 1, 1000)
for mode in ("sorted", "grid"):
    h = _lib.Handle(0)
    h.set_option("sort_points", 1)
    h.set_problem(ndim=2, xs=coords[0], ys=coords[1], zs=None, values=values, model_id=_lib.MODEL_IDS[cfg["model"]],
                  params=internal_params(cfg["model"], cfg["params"]))
    for _ in range(3):
        if mode == "sorted":
            h.set_points(px, py, None)
        else:
            h.set_grid((gx, gx))
        h.predict_moving_window(10)
    t = h.timing()
    print(mode, "search+rhs %.3f solve %.3f sorted %d" % (t["rhs_ms"], t["contract_ms"], t["points_sorted"]), flush=True)
    h.close()
