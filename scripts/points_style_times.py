"""Wall-clock of each stage of one execute('points') against execute('grid') on the same points (round 5: where do the 8 % of config 5 go?)."""
import os, sys, time
os.environ["MIK_FACTOR_CACHE"] = "0"
import numpy as np
sys.path.insert(0, ".")
from bench import CONFIGS, synth, make_model, grid_axes, shard_points

for c in (int(a) for a in (sys.argv[1:] or ["5", "2"])):
    cfg = CONFIGS[c]
    coords, values = synth(cfg["seed"], cfg["n"], cfg["ndim"])
    m = make_model(cfg, coords, values)
    axes = grid_axes(cfg, 1)
    parts = shard_points(cfg, 0, 1)
    h = m._get_handle()
    for style, args in (("grid", axes), ("points", parts)):
        best = None
        for rep in range(4):
            t = [time.perf_counter()]
            m._set_problem(h); t.append(time.perf_counter())
            h.factor(); t.append(time.perf_counter())
            P = m._prepare(style, args, None); t.append(time.perf_counter())
            P.load(h, cfg["ndim"]); t.append(time.perf_counter())
            h.predict(); t.append(time.perf_counter())
            z, ss = h.get_results(); t.append(time.perf_counter())
            d = [(b - a) * 1e3 for a, b in zip(t, t[1:])]
            t0 = time.perf_counter(); m.execute(style, *args, backend="loop"); e = (time.perf_counter() - t0) * 1e3
            if best is None or e < best[0]:
                best = (e, d)
        e, d = best
        print("config %d %-6s: execute %.2f ms | staged: set_problem %.2f factor %.2f prepare %.2f load %.2f predict %.2f get_results %.2f = %.2f  [ms]" % (
            c, style, e, d[0], d[1], d[2], d[3], d[4], d[5], sum(d)), flush=True)
