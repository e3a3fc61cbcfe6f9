"""Where the wall-clock of the class-level execute('grid') goes: every Handle method timed inside real execute() calls of a
bench config, in a loop that keeps the previous result alive like bench.py does."""
import os, sys, time, collections
os.environ["MIK_FACTOR_CACHE"] = "0"
import numpy as np
sys.path.insert(0, ".")
from bench import CONFIGS, synth, make_model, grid_axes
from pykrige_amd import _lib

acc = collections.OrderedDict()
def wrap(name):
    f = getattr(_lib.Handle, name)
    def g(self, *a, **k):
        t0 = time.perf_counter()
        try:
            return f(self, *a, **k)
        finally:
            acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
    setattr(_lib.Handle, name, g)
for n in ("set_problem", "factor", "set_grid", "predict", "timing", "get_results"):
    wrap(n)
for c in (int(a) for a in (sys.argv[1:] or ["2", "3"])):
    cfg = CONFIGS[c]
    coords, values = synth(cfg["seed"], cfg["n"], cfg["ndim"])
    m = make_model(cfg, coords, values)
    axes = grid_axes(cfg, 1)
    res = m.execute("grid", *axes, backend="loop")
    res = m.execute("grid", *axes, backend="loop")
    acc.clear()
    K = 4
    t0 = time.perf_counter()
    for _ in range(K):
        res = m.execute("grid", *axes, backend="loop")
    dt = (time.perf_counter() - t0) / K * 1e3
    parts = {k: v / K * 1e3 for k, v in acc.items()}
    print("config %d: execute %.3f ms per call; inside Handle calls %.3f ms (%s); Python around them %.3f ms" % (
        c, dt, sum(parts.values()), ", ".join("%s %.3f" % kv for kv in parts.items()), dt - sum(parts.values())), flush=True)
