"""A/B of the block sweep's trailing-update kernel: 4 waves per 128 x 128 tile (wave tile 64 x 64, the round-1/2 form) against
8 waves (32 x 64, 128 VGPRs, 4 waves per SIMD), option update_waves.  invert_ms (best of 5) per config, inverses compared bit for bit."""
import sys
import numpy as np
sys.path.insert(0, ".")
from bench import CONFIGS, synth, internal_params
from pykrige_amd import _lib

for cfgid in (3, 4, 2, 5):
    cfg = CONFIGS[cfgid]
    nd = cfg["ndim"]
    coords, values = synth(cfg["seed"], cfg["n"], nd)
    line, ref = "config %d N=%d:" % (cfgid, cfg["n"]), None
    for sym in (-1, 0):
        for uw in (4, 8):
            h = _lib.Handle(0)
            h.set_option("update_waves", uw)
            h.set_option("symsweep", sym)
            h.set_problem(ndim=nd, xs=coords[0], ys=coords[1], zs=coords[2] if nd == 3 else None, values=values,
                          model_id=_lib.MODEL_IDS[cfg["model"]], params=internal_params(cfg["model"], cfg["params"]),
                          regional_linear=bool(cfg.get("rl")), wells=np.array(cfg["wells"]) if cfg.get("wells") else None)
            h.factor()
            ts = []
            for _ in range(5):
                h.factor()
                ts.append(h.timing()["invert_ms"])
            a = h.get_matrix(1)
            same = ""
            if uw == 4:
                ref = a
            else:
                same = " (bit-identical: %s)" % bool(np.array_equal(a, ref))
            line += "  sweep %s / %d waves %.2f ms%s" % ("auto" if sym < 0 else "full", uw, min(ts), same)
            h.close()
    print(line, flush=True)
