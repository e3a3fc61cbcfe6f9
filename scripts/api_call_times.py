"""Wall-clock of each C-ABI call of one execute('grid') against the device phases it reports (where do the host-side gaps sit?)."""
import os, sys, time
os.environ["MIK_FACTOR_CACHE"] = "0"
import numpy as np
sys.path.insert(0, ".")
from bench import CONFIGS, synth, make_model, grid_axes

for c in (int(a) for a in (sys.argv[1:] or ["2", "3"])):
    cfg = CONFIGS[c]
    coords, values = synth(cfg["seed"], cfg["n"], cfg["ndim"])
    m = make_model(cfg, coords, values)
    axes = grid_axes(cfg, 1)
    h = m._get_handle()
    for rep in range(3):
        t = [time.perf_counter()]
        m._set_problem(h); t.append(time.perf_counter())
        h.factor(); t.append(time.perf_counter())
        P = m._prepare("grid", axes, None); t.append(time.perf_counter())
        P.load(h, cfg["ndim"]); t.append(time.perf_counter())
        h.predict(); t.append(time.perf_counter())
        tm = h.timing(); t.append(time.perf_counter())
        z, ss = h.get_results(); t.append(time.perf_counter())
        d = [(b - a) * 1e3 for a, b in zip(t, t[1:])]
    print("config %d: set_problem %.3f | factor %.3f (device: assemble %.3f + invert %.3f + verify %.3f) | prepare %.3f | set_grid %.3f | "
          "predict %.3f (device %.3f) | timing %.3f | get_results %.3f  [ms]" % (c, d[0], d[1], tm["assemble_ms"], tm["invert_ms"], tm["verify_ms"],
                                                                            d[2], d[3], d[4], tm["predict_ms"], d[5], d[6]), flush=True)
