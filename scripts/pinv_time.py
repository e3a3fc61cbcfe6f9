import sys, time
import numpy as np
sys.path.insert(0, ".")
import pykrige_amd as pa
from bench import synth
for n in (500, 1000, 2000, 4000):
    (x, y), v = synth(7, n, 2)
    x[-4:], y[-4:] = x[:4], y[:4]
    ok = pa.OrdinaryKriging(x, y, v, variogram_model="exponential", variogram_parameters=[1.0, 0.3, 0.0], pseudo_inv=True)
    t0 = time.perf_counter()
    z, ss = ok.execute("grid", np.linspace(0, 1, 50), np.linspace(0, 1, 50), backend="loop")
    print(n, "execute %.2f s" % (time.perf_counter() - t0), "invert_ms", ok.last_timing["invert_ms"], flush=True)
