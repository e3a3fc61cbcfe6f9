"""Gate 2 of the "K2 in a GEMM-rich form" question (GPU box): would a recursive Schur-complement inverse -- A11^-1 by the sweep, W = A21 A11^-1 and
S = A22 - W A12 as K = M/2 products of the contraction's tile engine, S^-1 by the sweep, B12 = -W^T S^-1, B11 = A11^-1 - B12 W -- beat the block
Gauss-Jordan sweep over the whole matrix?  Its cost is 2 sweeps of M/2 + 2 full and 2 symmetric M/2-cubed products (6 (M/2)^3 multiply-adds x 2,
the same M^3 flops as the half sweep).  Measured here: the sweep at M/2 and M (mik_timing invert_ms, best of 5) for the two shapes of the bench;
tools/kernel_bench (same box, same call) gives the products' rates at M/2."""
import sys
import numpy as np
sys.path.insert(0, ".")
from bench import synth, internal_params
from pykrige_amd import _lib

h = _lib.Handle(0)
print("%8s %6s %12s %10s %14s" % ("N", "Mp", "model", "invert ms", "TFLOP/s (M^3)"))
for model, params in (("exponential", [1.0, 0.3, 0.02]), ("spherical", [1.0, 0.3, 0.02])):
    for n in (2047, 2559, 4031, 4095, 5000, 8000):
        coords, values = synth(n, n, 2)
        best = None
        for _ in range(5):
            h.set_problem(ndim=2, xs=coords[0], ys=coords[1], zs=None, values=values, model_id=_lib.MODEL_IDS[model], params=internal_params(model, params))
            h.factor()
            t = h.timing()["invert_ms"]
            best = t if best is None else min(best, t)
        mp = -(-(n + 1) // 128) * 128
        print("%8d %6d %12s %10.3f %14.1f" % (n, mp, model, best, mp ** 3 / best * 1e-9), flush=True)
