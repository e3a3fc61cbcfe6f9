"""The general pseudo-inverse (factor_path 4, pinv_fast = 0): block one-sided Jacobi (option pinv_block = 1, round 4) against the scalar
form (0): time of mik_factor and agreement of the two results and with scipy.linalg.pinv.  Duplicated stations, zero nugget.  GPU box."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from bench import synth, internal_params
from pykrige_amd import _lib

for n in (500, 1000, 2000, 4000):
    (x, y), v = synth(n, n, 2)
    x[-8:], y[-8:] = x[:8], y[:8]
    res = {}
    for block in (1, 0):
        if not block and n > 2000 and "--all" not in sys.argv:
            continue
        h = _lib.Handle(0)
        h.set_option("pinv_fast", 0)
        h.set_option("pinv_block", block)
        h.set_problem(ndim=2, xs=x, ys=y, zs=None, values=v, model_id=_lib.MODEL_IDS["exponential"],
                      params=internal_params("exponential", [1.0, 0.3, 0.0]), pseudo_inv=1)
        t0 = time.perf_counter()
        try:
            h.factor()
        except Exception as e:  # noqa: BLE001
            print("M=%5d  pinv_block=%d: %r" % (n + 1, block, e), flush=True)
            h.close()
            continue
        res[block] = (time.perf_counter() - t0, h.timing()["factor_path"], h.get_matrix(1))
        h.close()
    line = "M=%5d  block Jacobi %9.1f ms (path %d)" % (n + 1, res[1][0] * 1e3, res[1][1])
    if 0 in res:
        line += "   scalar Jacobi %9.1f ms   max|diff| / max|pinv| %.1e" % (res[0][0] * 1e3, np.abs(res[1][2] - res[0][2]).max() / np.abs(res[0][2]).max())
    if n <= 2000:
        import scipy.linalg
        from oracle import kriging_oracle as ko
        st = ko.KrigingState(ndim=2, coords_orig=np.stack([x, y], 1), values=v, model="exponential", params=ko.internal_parameters("exponential", [1.0, 0.3, 0.0]))
        ref = scipy.linalg.pinv(ko.kriging_matrix(st))
        line += "   vs scipy.linalg.pinv %.1e" % (np.abs(res[1][2] - ref).max() / np.abs(ref).max())
    print(line, flush=True)
