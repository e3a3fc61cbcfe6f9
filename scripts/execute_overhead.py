"""End-to-end OrdinaryKriging.execute('grid') wall time at config 2 vs the library's own predict time: what the host-side
Python (meshgrid, anisotropy adjustment, H2D / D2H, reshapes) adds."""
import cProfile, os, pstats, sys, time
os.environ["MIK_FACTOR_CACHE"] = "0"  # every execute() assembles and inverts, as the reference does
import numpy as np
sys.path.insert(0, ".")
from bench import CONFIGS, synth
import pykrige_amd as pa

cfg = CONFIGS[int(sys.argv[1]) if len(sys.argv) > 1 else 2]
coords, values = synth(cfg["seed"], cfg["n"], cfg["ndim"])
axes = [np.linspace(0, 1, n) for n in cfg["grid"]]
if cfg["ndim"] == 2:
    m = pa.OrdinaryKriging(coords[0], coords[1], values, variogram_model=cfg["model"], variogram_parameters=cfg["params"])
else:
    m = pa.OrdinaryKriging3D(coords[0], coords[1], coords[2], values, variogram_model=cfg["model"], variogram_parameters=cfg["params"])
for backend in ("vectorized", "loop"):
    m.execute("grid", *axes, backend=backend)
    t0 = time.perf_counter()
    z, ss = m.execute("grid", *axes, backend=backend)
    dt = time.perf_counter() - t0
    t = m.last_timing
    dev = t["assemble_ms"] + t["invert_ms"] + t["predict_ms"]
    print("backend=%-10s execute() %.1f ms   device %.1f ms   host overhead %.1f ms (%.1f %%)" % (backend, dt * 1e3, dev, dt * 1e3 - dev, 100 * (dt * 1e3 - dev) / (dt * 1e3)))
pr = cProfile.Profile()
pr.enable()
m.execute("grid", *axes, backend="loop")
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
