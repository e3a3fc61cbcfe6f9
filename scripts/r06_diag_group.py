"""diagnostic (round 6): where does a 2-member aliased group differ from one device at N = 1300?  (z differs by 3e-15: the factor?)"""
import sys
import numpy as np
sys.path.insert(0, ".")
from pykrige_amd import _lib as lib
from tests import _fixtures as fx

rng = np.random.default_rng(5)
pts = [rng.random(6000), rng.random(6000)]


def run(h, c, v, model, params):
    h.set_problem(ndim=2, xs=c[0], ys=c[1], zs=None, values=v, model_id=lib.MODEL_IDS[model], params=params)
    h.set_points(pts[0], pts[1])
    h.factor()
    h.predict()
    z, s = h.get_results()
    return z.copy(), s.copy(), h.get_matrix(1), dict(h.timing())


for n in (700, 1300):
    c, v = fx.synth(21, n, 2)
    for tri in (1, 0):
        for chunk in (None, 1024):
            outs = []
            for members in (1, 1, 2):
                h = lib.Handle(0)
                if members > 1:
                    h.set_devices(members, alias=True)
                    h.set_option("exchange_tri", tri)
                if chunk:
                    h.set_option("chunk", chunk)
                outs.append(run(h, c, v, "exponential", [0.9, 0.3, 0.1]))
                h.close()
            a, b, g = outs
            half = 3072
            print("n %d tri %d chunk %s: single vs single dz %.2e dA %.2e | group vs single: dz leader slab %.2e, member slab %.2e, dss %.2e / %.2e, dA(leader) %.2e; path %d attempts %d/%d half %d/%d" % (
                n, tri, chunk, np.abs(a[0] - b[0]).max(), np.abs(a[2] - b[2]).max(), np.abs(g[0][:half] - a[0][:half]).max(), np.abs(g[0][half:] - a[0][half:]).max(),
                np.abs(g[1][:half] - a[1][:half]).max(), np.abs(g[1][half:] - a[1][half:]).max(), np.abs(g[2] - a[2]).max(), g[3]["exchange_path"],
                a[3]["factor_attempts"], g[3]["factor_attempts"], a[3]["half_sweep"], g[3]["half_sweep"]), flush=True)
