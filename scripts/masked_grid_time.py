"""execute('masked') on a large grid with few stations: the host-side O(npt) work (index list of the unmasked cells, gathers,
scatter of the results) against the device time.  N=500 stations, 4096 x 4096 grid, 35 % of the cells masked."""
import os, sys, time
import numpy as np
sys.path.insert(0, ".")
import pykrige_amd as pa
from bench import synth

(x, y), v = synth(9, 500, 2)
ok = pa.OrdinaryKriging(x, y, v, variogram_model="exponential", variogram_parameters=[1.0, 0.3, 0.02])
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ax = np.linspace(0, 1, n)
rng = np.random.default_rng(0)
mask = rng.random((n, n)) < 0.35
for style, kw in (("grid", {}), ("masked", {"mask": mask})):
    ok.execute(style, ax, ax, backend="loop", **kw)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        z, ss = ok.execute(style, ax, ax, backend="loop", **kw)
        ts.append(time.perf_counter() - t0)
    t = ok.last_timing
    dev = t["predict_ms"]  # the factorization is the cached one (same stations, same model)
    print("N=500, %d x %d, style=%-6s: execute() %.1f ms, of which mik_predict on the device %.1f ms, the rest %.1f ms" % (n, n, style, ts[-1] * 1e3, dev, ts[-1] * 1e3 - dev), flush=True)
