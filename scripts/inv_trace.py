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
h.factor(); h.factor()
print(h.timing()["invert_ms"])
