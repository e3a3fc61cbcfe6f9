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
    for la, early, gate in ((0, 0, -1), (1, 0, -1), (1, 1, -1), (1, 5, -1), (1, 4, -1)):
        for diag, sym in ((1, 0), (1, 1)):
            h = _lib.Handle(0)
            h.set_option("lookahead", la)
            h.set_option("early_diag", early)
            h.set_option("gate", gate)
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
            line += "  la%d%s/d%d/sym%d %.2f ms (%.0e)" % (la, ("", "e", "e2", "e3", "e4", "e5")[early] + ("" if gate < 0 else "g%d" % gate), diag, sym, min(ts), np.abs(a - ref).max() / np.abs(ref).max())
            h.close()
    print(line, flush=True)

# where does the look-ahead (early-diagonal schedule) start to pay?  config-3-like stations, growing N
for n in (300, 500, 700, 900, 1200, 1500):
    coords, values = synth(3, n, 2)
    line = "N=%d (%d block columns):" % (n, (n + 1 + 127) // 128)
    for la in (0, 1):
        h = _lib.Handle(0)
        h.set_option("lookahead", la)
        h.set_option("factor", 1)
        h.set_problem(ndim=2, xs=coords[0], ys=coords[1], zs=None, values=values, model_id=_lib.MODEL_IDS["exponential"],
                      params=internal_params("exponential", [1.0, 0.3, 0.0]))
        h.factor()
        ts = []
        for _ in range(6):
            h.factor()
            ts.append(h.timing()["invert_ms"])
        line += "  la%d %.3f ms" % (la, min(ts))
        h.close()
    print(line, flush=True)
