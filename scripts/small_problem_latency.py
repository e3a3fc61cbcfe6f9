"""execute("grid") wall time for small problems (where launch overheads, not flops, decide) against the NumPy/SciPy
restatement of backend='vectorized' on the same host."""
import os, sys, time
import numpy as np
sys.path.insert(0, ".")
import pykrige_amd as pa
from bench import synth
from oracle import kriging_oracle as ko

for n, g in ((100, 50), (500, 100), (1000, 200), (2000, 300), (5000, 256)):
    (x, y), v = synth(n, n, 2)
    ax = np.linspace(0, 1, g)
    user = [1.0, 0.3, 0.02]
    ok = pa.OrdinaryKriging(x, y, v, variogram_model="exponential", variogram_parameters=user)
    ok.execute("grid", ax, ax, backend="loop")
    ts, tc_ = [], []
    os.environ["MIK_FACTOR_CACHE"] = "0"  # every call assembles and inverts, as the reference does
    t = None
    for _ in range(5):
        t0 = time.perf_counter()
        ok.execute("grid", ax, ax, backend="loop")
        ts.append(time.perf_counter() - t0)
        if ts[-1] == min(ts):
            t = ok.last_timing  # device phases of the fastest call
    os.environ["MIK_FACTOR_CACHE"] = "1"  # the factored matrix stays on the device while the problem is unchanged
    ok.execute("grid", ax, ax, backend="loop")
    for _ in range(5):
        t0 = time.perf_counter()
        ok.execute("grid", ax, ax, backend="loop")
        tc_.append(time.perf_counter() - t0)
    st = ko.KrigingState(ndim=2, coords_orig=np.stack([x, y], 1), values=v, model="exponential",
                         params=ko.internal_parameters("exponential", user))
    t0 = time.perf_counter()
    ko.execute(st, "grid", ax, ax)
    tc = time.perf_counter() - t0
    print("N=%5d grid %3dx%-3d  GPU execute() %8.2f ms (device: invert %.2f + predict %.2f); repeated on the same object (cached factor) %8.2f ms   CPU vectorized %9.1f ms   x%.0f" % (
        n, g, g, min(ts) * 1e3, t["invert_ms"], t["predict_ms"], min(tc_) * 1e3, tc * 1e3, tc / min(ts)), flush=True)
