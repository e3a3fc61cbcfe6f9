"""Differential run on the GPU box: the objects the real reference (oracle/_ref) and the drop-in construct from the same arguments -- every attribute of the
reference's instance (name, type, dtype, shape, value) and what get_variogram_points / get_epsilon_residuals / get_statistics return.  Exits non-zero on a difference."""
import sys, numpy as np
sys.path.insert(0, ".")
import pykrige_amd as pa
from oracle import ref_package as rp
pk = rp.import_reference(stub_statistics=False)
rng = np.random.default_rng(7); n = 40
x, y, z3 = rng.random(n), rng.random(n), rng.random(n); v = np.sin(4 * x) + y + 0.1 * rng.standard_normal(n)
BAD = []


def cmp(name, a, b):
    da, db = a.__dict__, {k: getattr(b, k) for k in a.__dict__ if hasattr(b, k)}
    missing = [k for k in da if not hasattr(b, k)]
    bad = []
    for k, va in da.items():
        if k in missing: continue
        vb = db[k]
        if callable(va) and callable(vb): continue
        try:
            if isinstance(va, np.ndarray) or isinstance(vb, np.ndarray) or isinstance(va, (float, np.floating)):
                aa, bb = np.asarray(va, dtype=float), np.asarray(vb, dtype=float)
                if aa.shape != bb.shape: bad.append("%s shape %s vs %s" % (k, aa.shape, bb.shape))
                elif aa.size and not np.allclose(aa, bb, rtol=1e-6, atol=1e-9, equal_nan=True): bad.append("%s max|d| %.2e" % (k, np.nanmax(np.abs(aa - bb))))
                if isinstance(va, np.ndarray) and isinstance(vb, np.ndarray) and va.dtype != vb.dtype: bad.append("%s dtype %s vs %s" % (k, va.dtype, vb.dtype))
                if type(va) is not type(vb) and not (isinstance(va, (float, np.floating)) and isinstance(vb, (float, np.floating))): bad.append("%s type %s vs %s" % (k, type(va).__name__, type(vb).__name__))
            elif isinstance(va, (list, tuple)):
                if len(va) != len(vb): bad.append("%s len" % k)
                elif len(va) and all(isinstance(q, (int, float, np.floating)) for q in va) and not np.allclose(np.asarray(va, float), np.asarray(vb, float), rtol=1e-6, atol=1e-9): bad.append("%s values %s vs %s" % (k, va, vb))
                if type(va) is not type(vb): bad.append("%s type %s vs %s" % (k, type(va).__name__, type(vb).__name__))
            elif va != vb and not (va is None and vb is None):
                bad.append("%s %r vs %r" % (k, va, vb))
        except Exception as e:
            bad.append("%s compare failed: %s" % (k, e))
    print("%-40s missing %s; differing %s" % (name, missing, bad))
    if missing or bad:
        BAD.append(name)
cases = {
 "OK fitted linear": lambda m: m.ok.OrdinaryKriging(x, y, v),
 "OK fitted spherical stats": lambda m: m.ok.OrdinaryKriging(x, y, v, variogram_model="spherical", nlags=8, weight=True, enable_statistics=True),
 "OK given params geographic": lambda m: m.ok.OrdinaryKriging(x * 50, y * 50, v, variogram_model="exponential", variogram_parameters=[1., 20., .1], coordinates_type="geographic"),
 "OK anisotropy": lambda m: m.ok.OrdinaryKriging(x, y, v, variogram_model="gaussian", anisotropy_scaling=2., anisotropy_angle=30.),
 "UK regional+point_log": lambda m: m.uk.UniversalKriging(x, y, v, variogram_model="linear", drift_terms=["regional_linear", "point_log"], point_drift=[[.2, .3, 1.]]),
 "UK external": lambda m: m.uk.UniversalKriging(x, y, v, variogram_model="power", drift_terms=["external_Z"], external_drift=rng.random((5, 6)) * 0 + np.arange(30).reshape(5, 6), external_drift_x=np.linspace(-.1, 1.1, 6), external_drift_y=np.linspace(-.1, 1.1, 5)),
 "OK3D": lambda m: m.ok3d.OrdinaryKriging3D(x, y, z3, v, variogram_model="exponential", anisotropy_scaling_y=2., anisotropy_angle_z=20.),
 "UK3D": lambda m: m.uk3d.UniversalKriging3D(x, y, z3, v, variogram_model="spherical", drift_terms=["regional_linear"]),
}
for name, mk in cases.items():
    try:
        a = mk(pk)
    except Exception as e:
        print(name, "reference raised", type(e).__name__, e); continue
    try:
        b = mk(pa)
    except Exception as e:
        print(name, "DROP-IN raised", type(e).__name__, e); BAD.append(name); continue
    cmp(name, a, b)
    for meth in ("get_variogram_points", "get_epsilon_residuals", "get_statistics"):
        if hasattr(a, meth):
            try: ra = getattr(a, meth)()
            except Exception as e: ra = e
            try: rb = getattr(b, meth)()
            except Exception as e: rb = e
            if isinstance(ra, Exception) or isinstance(rb, Exception):
                print("   %s: %s / %s" % (meth, type(ra).__name__, type(rb).__name__))
                if type(ra) is not type(rb):
                    BAD.append(name + "." + meth)
            else:
                fa = np.concatenate([np.ravel(np.asarray(q, float)) for q in (ra if isinstance(ra, tuple) else (ra,))])
                fb = np.concatenate([np.ravel(np.asarray(q, float)) for q in (rb if isinstance(rb, tuple) else (rb,))])
                same = fa.shape == fb.shape and np.allclose(fa, fb, rtol=1e-6, atol=1e-9, equal_nan=True)
                print("   %s: %s" % (meth, "same" if same else "DIFFER %s %s" % (fa[:4], fb[:4])))
                if not same:
                    BAD.append(name + "." + meth)
print("%d difference(s)%s" % (len(BAD), ": " + "; ".join(BAD) if BAD else ""))
sys.exit(1 if BAD else 0)
