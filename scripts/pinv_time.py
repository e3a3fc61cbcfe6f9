"""pseudo_inv=True on duplicated stations: the deflated regular inverse (factor_path 5) against the Jacobi pseudo-inverse
(pinv_fast = 0, factor_path 4), time of mik_factor and agreement of the two inverses."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from bench import synth, internal_params
from pykrige_amd import _lib

for n in (500, 1000, 2000, 4000):
    (x, y), v = synth(n, n, 2)
    x[-8:], y[-8:] = x[:8], y[:8]
    res = {}
    for fast in (1, 0):
        if not fast and n > 2000:
            continue
        h = _lib.Handle(0)
        h.set_option("pinv_fast", fast)
        h.set_problem(ndim=2, xs=x, ys=y, zs=None, values=v, model_id=_lib.MODEL_IDS["exponential"],
                      params=internal_params("exponential", [1.0, 0.3, 0.0]), pseudo_inv=1)
        h.factor()
        t0 = time.perf_counter()
        h.factor()
        res[fast] = (time.perf_counter() - t0, h.timing()["factor_path"], h.get_matrix(1))
        h.close()
    line = "N=%5d  deflated inverse %8.2f ms (path %d)" % (n, res[1][0] * 1e3, res[1][1])
    if 0 in res:
        line += "   Jacobi %9.1f ms (path %d)   max|diff| / max|pinv| %.1e" % (res[0][0] * 1e3, res[0][1], np.abs(res[1][2] - res[0][2]).max() / np.abs(res[0][2]).max())
    print(line, flush=True)

# a rank deficiency that is not duplicated stations: collinear stations under a regional-linear drift (null space 1) -- the
# numerically deflated inverse (factor_path 6, round 3) against the Jacobi pseudo-inverse
for n in (500, 1000, 2000, 4000):
    rng = np.random.default_rng(n)
    x = rng.random(n)
    y = 0.5 * x + 0.2
    v = np.sin(5 * x) + 0.1 * rng.standard_normal(n)
    res = {}
    for fast in (1, 0):
        if not fast and n > 1000:
            continue
        h = _lib.Handle(0)
        h.set_option("pinv_fast", fast)
        h.set_problem(ndim=2, xs=x, ys=y, zs=None, values=v, model_id=_lib.MODEL_IDS["exponential"],
                      params=internal_params("exponential", [1.0, 0.3, 0.01]), regional_linear=True, pseudo_inv=1)
        h.factor()
        t0 = time.perf_counter()
        h.factor()
        t = h.timing()
        res[fast] = (time.perf_counter() - t0, t["factor_path"], t["null_dim"], h.get_matrix(1))
        h.close()
    line = "collinear + regional_linear  N=%5d  fast path %8.2f ms (path %d, null space %d)" % (n, res[1][0] * 1e3, res[1][1], res[1][2])
    if 0 in res:
        line += "   Jacobi %9.1f ms (path %d)   max|diff| / max|pinv| %.1e" % (res[0][0] * 1e3, res[0][1], np.abs(res[1][3] - res[0][3]).max() / np.abs(res[0][3]).max())
    print(line, flush=True)
