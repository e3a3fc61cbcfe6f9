"""Moving-window kriging on config-2 stations for a few window sizes (one call each) -- run under rocprofv3 --kernel-trace."""
import sys
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
npt = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
h.set_points(rng.random(npt), rng.random(npt), None)
for k in (8, 15, 16, 31, 32, 50, 63, 64, 100):
    h.predict_moving_window(k)
    print(k, h.timing()["predict_ms"], flush=True)
