"""The reference's own benchmark (/root/reference/benchmarks/kriging_benchmarks.py:11-13,118: OrdinaryKriging 2-D, linear
variogram, N_train in {400, 800}, 1000-2000 random test points, backends x moving windows None / 10 / 50 / 100), timed
side by side: pykrige_amd on the GPU against the reference's compiled C loops (oracle/_ref: lib/cok.pyx _c_exec_loop and
_c_exec_loop_moving_window, kind "reference") and the NumPy restatement of backend='vectorized' (kind "port") on the
host.  Same random data as the reference script (np.random.seed(19999)); the variogram is fitted by the class under test
and the SAME fitted parameters are handed to the CPU legs.  Seconds per execute("points", ...) call, best of 3.

    python scripts/reference_benchmark_shapes.py > profiles/r02_reference_benchmark_shapes.txt
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pykrige_amd as pa  # noqa: E402
from oracle import kriging_oracle as ko  # noqa: E402
from oracle import ref_c_loop as rc  # noqa: E402

WINDOWS = [None, 10, 50, 100]


def best_of(fn, k=3):
    best, out = None, None
    for _ in range(k):
        t0 = time.perf_counter()
        out = fn()
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    return best, out


def main():
    np.random.seed(19999)
    print("%-22s %-8s %12s %12s %12s %10s %12s %12s" % ("shape", "window", "gpu 'C' s", "ref C s", "port vec s", "C / gpu", "max|dz|", "max|dss|"))
    for n_train, n_test in [(400, 1000), (400, 2000), (800, 2000)]:
        X = np.random.rand(n_train, 2)
        y = np.random.rand(n_train)
        P = np.random.rand(n_test, 2)
        t0 = time.perf_counter()
        ok = pa.OrdinaryKriging(X[:, 0], X[:, 1], y, variogram_model="linear", verbose=False, enable_plotting=False)
        t_train = time.perf_counter() - t0
        st = ko.KrigingState(ndim=2, coords_orig=X, values=y, model="linear", params=list(ok.variogram_model_parameters))
        ok.execute("points", P[:, 0], P[:, 1], backend="C")  # warm: library load, buffers
        for w in WINDOWS:
            kw = {} if w is None else {"n_closest_points": w}
            os.environ["MIK_FACTOR_CACHE"] = "0"  # the reference re-inverts on every call; so does this timing
            tg, (zg, sg) = best_of(lambda: ok.execute("points", P[:, 0], P[:, 1], backend="C", **kw))
            if w is None:
                tc, (zc, sc, _) = best_of(lambda: rc.c_backend(st, P))
                tv, _ = best_of(lambda: ko.solve_points(st, P))
            else:
                tc, (zc, sc, _, _) = best_of(lambda: rc.c_backend_moving_window(st, P, w))
                tv = float("nan")  # the reference has no vectorized moving window (ok.py:982-986)
            print("%-22s %-8s %12.5f %12.5f %12.5f %10.1f %12.2e %12.2e" % (
                "N=%d, %d points" % (n_train, n_test), w, tg, tc, tv, tc / tg, np.abs(zg - zc).max(), np.abs(sg - sc).max()))
        print("%-22s training (variogram fit in the constructor): %.3f s" % ("N=%d" % n_train, t_train))


if __name__ == "__main__":
    main()
