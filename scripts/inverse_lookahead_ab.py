"""A/B of the block-sweep inverse variants on the bench configs: look-ahead on/off x diagonal-block kernel variant x
symmetric (lower-triangle) update; the relative deviation from the first variant is printed in parentheses."""
import sys
import numpy as np
sys.path.insert(0, ".")
from bench import CONFIGS, synth, internal_params
from pykrige_amd import _lib

for cfgid in (2, 3, 5):
    cfg = CONFIGS[cfgid]
    nd = cfg["ndim"]
    coords, values = synth(cfg["seed"], cfg["n"], nd)
    ref = None
    line = "config %d N=%d:" % (cfgid, cfg["n"])
    for la in (0, 1):
        for diag, sym in ((0, 0), (1, 0), (1, 1)):
            h = _lib.Handle(0)
            h.set_option("lookahead", la)
            h.set_option("diag", diag)
            h.set_option("symsweep", sym)
            h.set_problem(ndim=nd, xs=coords[0], ys=coords[1], zs=coords[2] if nd == 3 else None, values=values,
                          model_id=_lib.MODEL_IDS[cfg["model"]], params=internal_params(cfg["model"], cfg["params"]))
            h.factor()
            ts = []
            for _ in range(4):
                h.factor()
                ts.append(h.timing()["invert_ms"])
            a = h.get_matrix(1)
            if ref is None:
                ref = a
            line += "  la%d/d%d/sym%d %.2f ms (%.0e)" % (la, diag, sym, min(ts), np.abs(a - ref).max() / np.abs(ref).max())
            h.close()
    print(line, flush=True)
