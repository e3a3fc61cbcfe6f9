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
for n in ("set_problem", "factor", "set_grid", "predict", "predict_moving_window", "timing", "get_results", "take_results"):
    if hasattr(_lib.Handle, n):
        wrap(n)
# `--window K`: the moving window (n_closest_points = K) instead of the dense path
window = int(sys.argv[sys.argv.index("--window") + 1]) if "--window" in sys.argv else None
args = [a for a in sys.argv[1:] if a != "--window" and not (window is not None and a == str(window) and sys.argv[sys.argv.index(a) - 1] == "--window")]
kw = dict(backend="loop", **({"n_closest_points": window} if window else {}))
for c in (int(a) for a in (args or ["2", "3"])):
    cfg = CONFIGS[c]
    coords, values = synth(cfg["seed"], cfg["n"], cfg["ndim"])
    m = make_model(cfg, coords, values)
    axes = grid_axes(cfg, 1)
    res = m.execute("grid", *axes, **kw)
    res = m.execute("grid", *axes, **kw)
    acc.clear()
    K = 8 if window else 4
    t0 = time.perf_counter()
    for _ in range(K):
        res = m.execute("grid", *axes, **kw)
    dt = (time.perf_counter() - t0) / K * 1e3
    parts = {k: v / K * 1e3 for k, v in acc.items()}
    if window:
        t = m.last_timing
        print("  moving window k = %d: device phases of the last call: knn+rhs %.3f ms, solve %.3f ms, predict_total %.3f ms" % (
            window, t.get("rhs_ms", 0.0), t.get("contract_ms", 0.0), t.get("predict_ms", 0.0)))
        import cProfile, pstats, io
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(K):
            res = m.execute("grid", *axes, **kw)
        pr.disable()
        st = io.StringIO()
        pstats.Stats(pr, stream=st).sort_stats("cumulative").print_stats(18)
        print("  cProfile of %d more calls (cumulative):" % K)
        print("\n".join("  " + ln for ln in st.getvalue().splitlines()[4:40]))
    print("config %d: execute %.3f ms per call; inside Handle calls %.3f ms (%s); Python around them %.3f ms" % (
        c, dt, sum(parts.values()), ", ".join("%s %.3f" % kv for kv in parts.items()), dt - sum(parts.values())), flush=True)
