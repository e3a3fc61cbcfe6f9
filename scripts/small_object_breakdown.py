import sys, time, numpy as np
sys.path.insert(0, ".")
import pykrige_amd as pa
from pykrige_amd import _lib
rng = np.random.default_rng(1); gx = np.linspace(0, 1, 64)
pa.OrdinaryKriging(rng.random(50), rng.random(50), rng.random(50), variogram_model="exponential", variogram_parameters=[1., .3, .05]).execute("grid", gx, gx)
for n in (100, 500, 2000):
    tc, t1, t2, th = [], [], [], []
    for i in range(20):
        x, y, v = rng.random(n), rng.random(n), rng.random(n)
        t0 = time.perf_counter(); m = pa.OrdinaryKriging(x, y, v, variogram_model="exponential", variogram_parameters=[1., .3, .05]); tc.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); hh = m._get_handle(); th.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); m.execute("grid", gx, gx); t1.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); m.execute("grid", gx, gx); t2.append(time.perf_counter() - t0)
        lt = m.last_timing
    f = lambda a: 1e3 * float(np.median(a))
    print("N=%5d: construct %.2f ms, handle %.2f ms, first execute %.2f ms, second execute (cached factor) %.2f ms; device phases of the last call: %s" % (n, f(tc), f(th), f(t1), f(t2), {k: round(val, 3) for k, val in lt.items() if k.endswith("_ms") and val}))
import cProfile, pstats
x, y, v = rng.random(500), rng.random(500), rng.random(500)
pr = cProfile.Profile(); pr.enable()
for i in range(20):
    m = pa.OrdinaryKriging(x, y, v, variogram_model="exponential", variogram_parameters=[1., .3, .05]); m.execute("grid", gx, gx)
pr.disable(); pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
