"""A/B of one option of the block sweep (K2): invert_ms (best of 5) per bench config for each value of the option, inverses compared
bit for bit with the first value's.  usage: sweep_option_ab.py <option> <value> [<value> ...] [--configs 3,4,2,5]"""
import sys
import numpy as np
sys.path.insert(0, ".")
from bench import CONFIGS, synth, internal_params
from pykrige_amd import _lib

args = [a for a in sys.argv[1:] if not a.startswith("--")]
cfgs = [int(c) for a in sys.argv[1:] if a.startswith("--configs=") for c in a.split("=")[1].split(",")] or [3, 4, 2, 5]
opt, vals = args[0], [float(v) for v in args[1:]]
for cfgid in cfgs:
    cfg = CONFIGS[cfgid]
    nd = cfg["ndim"]
    coords, values = synth(cfg["seed"], cfg["n"], nd)
    line, ref = "config %d N=%d:" % (cfgid, cfg["n"]), None
    for val in vals:
        h = _lib.Handle(0)
        h.set_option(opt, val)
        for kv in [a for a in sys.argv[1:] if a.startswith("--set=")]:
            k, v = kv[6:].split("=")
            h.set_option(k, float(v))
        h.set_problem(ndim=nd, xs=coords[0], ys=coords[1], zs=coords[2] if nd == 3 else None, values=values,
                      model_id=_lib.MODEL_IDS[cfg["model"]], params=internal_params(cfg["model"], cfg["params"]),
                      regional_linear=bool(cfg.get("rl")), wells=np.array(cfg["wells"]) if cfg.get("wells") else None)
        h.factor()
        ts = []
        for _ in range(5):
            h.factor()
            ts.append(h.timing()["invert_ms"])
        a = h.get_matrix(1)
        if ref is None:
            ref, same = a, ""
        else:
            same = " (bit-identical: %s, max |diff| %.1e)" % (bool(np.array_equal(a, ref)), float(np.abs(a - ref).max()))
        line += "  %s=%g %.2f ms%s" % (opt, val, min(ts), same)
        h.close()
    print(line, flush=True)
