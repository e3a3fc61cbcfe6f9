"""Calibration of the inverse probe (mik_timing.verify_res_z / verify_res_inv, include/mikrige.h option "verify"): for the
randomized set-ups of tests/test_randomized_parity.py (dense cases) and for larger, deliberately ill-conditioned ones, every
factor path (half sweep, full sweep, pivoted) is run with the fallback disabled, and the probe's residuals are printed next
to the TRUE errors of z / sigma^2 against the CPU oracle.  Usage: python scripts/verify_probe_calibration.py [n_random] [big]"""
import sys, collections, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import importlib.util
spec = importlib.util.spec_from_file_location("t", "tests/test_randomized_parity.py")
m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
import pykrige_amd as pa
from oracle import kriging_oracle as ko

PATHS = (("half", 1, 1), ("full", 1, 0), ("pivoted", 2, 0))
rows = []


def run(tag, mdl, st, style, axes, mask, shape):
    zr, sr = ko.execute(st, style, *axes, mask=mask)
    keep = np.ones(shape, bool) if mask is None else ~mask
    h = mdl._get_handle()
    h.set_option("verify_tol_z", 1e300)
    h.set_option("verify_tol_inv", 1e300)
    for name, fac, half in PATHS:
        h.set_option("factor", fac)
        h.set_option("symsweep", half)
        try:
            z, ss = mdl.execute(style, *axes, mask=mask, backend="loop") if mask is not None else mdl.execute(style, *axes, backend="loop")
        except Exception as e:
            rows.append((tag, name, None, None, None, None, repr(e)[:60]))
            continue
        t = mdl.last_timing
        dz = float(np.abs(np.ma.getdata(z) - np.ma.getdata(zr))[keep].max())
        ds = float(np.abs(np.ma.getdata(ss) - np.ma.getdata(sr))[keep].max())
        rows.append((tag, name, t["verify_res_z"], t["verify_res_inv"], dz, ds, "attempts %d path %d half %d verify %.3f ms invert %.2f ms" % (
            t["factor_attempts"], t["factor_path"], t["half_sweep"], t["verify_ms"], t["invert_ms"])))


nrand = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for seed in range(nrand):
    c = m._case(seed)
    if c["window"] or c["drift"].get("specified") or c["drift"].get("functional") or c["geographic"]:
        continue
    nd, coords, values = c["ndim"], c["coords"], c["values"]
    st = ko.KrigingState(ndim=nd, coords_orig=coords, values=values, model=c["model"], params=ko.internal_parameters(c["model"], c["user"]),
                         scaling=c["scaling"], angle=c["angle"], exact_values=c["exact"],
                         regional_linear=bool(c["drift"].get("regional_linear")), point_log=c["drift"].get("wells"))
    kw = dict(variogram_model=c["model"], variogram_parameters=list(c["user"]), exact_values=c["exact"])
    if nd == 2:
        kw.update(anisotropy_scaling=c["scaling"][0], anisotropy_angle=c["angle"][0]); args = (coords[:, 0], coords[:, 1], values)
    else:
        kw.update(anisotropy_scaling_y=c["scaling"][0], anisotropy_scaling_z=c["scaling"][1], anisotropy_angle_x=c["angle"][0],
                  anisotropy_angle_y=c["angle"][1], anisotropy_angle_z=c["angle"][2]); args = (coords[:, 0], coords[:, 1], coords[:, 2], values)
    if c["universal"]:
        terms = (["regional_linear"] if c["drift"].get("regional_linear") else []) + (["point_log"] if "wells" in c["drift"] else [])
        if "wells" in c["drift"]:
            kw["point_drift"] = c["drift"]["wells"]
        mdl = (pa.UniversalKriging if nd == 2 else pa.UniversalKriging3D)(*args, drift_terms=terms, **kw)
    else:
        mdl = (pa.OrdinaryKriging if nd == 2 else pa.OrdinaryKriging3D)(*args, **kw)
    run("rand%d %s %s N=%d" % (seed, c["model"], "UK" if c["universal"] else "OK", c["n"]), mdl, st, c["style"], c["axes"], c["mask"], c["shape"])

if len(sys.argv) > 2:  # larger and deliberately ill-conditioned: long ranges, tiny nuggets, drift terms
    big = [("exponential", [1.0, 0.3, 0.0], 3200, False), ("exponential", [1.0, 3.0, 0.0], 3200, False), ("exponential", [1.0, 30.0, 0.0], 3200, False),
           ("exponential", [1.0, 3.0, 0.0], 3200, True), ("spherical", [1.0, 0.2, 0.01], 3200, False), ("spherical", [1.0, 2.0, 0.0], 3200, False),
           ("spherical", [1.0, 20.0, 0.0], 3200, True), ("gaussian", [1.0, 0.4, 0.02], 3200, False), ("gaussian", [1.0, 0.4, 0.001], 3200, False),
           ("linear", [1.0, 0.0], 3200, False), ("power", [1.0, 1.5, 0.0], 3200, True), ("power", [1.0, 1.9, 0.0], 3200, True),
           ("exponential", [1.0, 10.0, 0.0], 5000, False), ("spherical", [1.0, 5.0, 0.0], 5000, True)]
    for k, (model, user, n, uk) in enumerate(big):
        rng = np.random.default_rng(100 + k)
        x, y = rng.random(n), rng.random(n)
        v = np.sin(6 * x) * np.cos(4 * y) + 0.1 * rng.standard_normal(n)
        gx, gy = np.linspace(0, 1, 23), np.linspace(0, 1, 19)
        st = ko.KrigingState(ndim=2, coords_orig=np.stack([x, y], 1), values=v, model=model, params=ko.internal_parameters(model, user),
                             regional_linear=uk)
        if uk:
            mdl = pa.UniversalKriging(x, y, v, variogram_model=model, variogram_parameters=list(user), drift_terms=["regional_linear"])
        else:
            mdl = pa.OrdinaryKriging(x, y, v, variogram_model=model, variogram_parameters=list(user))
        t0 = time.time()
        run("big %s %s %s N=%d" % (model, user, "UK" if uk else "OK", n), mdl, st, "grid", (gx, gy), None, (19, 23))
        print("# %s done in %.0f s" % (model, time.time() - t0), flush=True)

print("%-44s %-8s %10s %10s %10s %10s  %s" % ("case", "path", "res_z", "res_inv", "true|dz|", "true|dss|", "info"))
for r in rows:
    if r[2] is None:
        print("%-44s %-8s %s" % (r[0], r[1], r[6]))
    elif r[0].startswith("big") or r[4] > 1e-10 or r[5] > 1e-9 or r[2] > 1e-11 or r[3] > 1e-10:
        print("%-44s %-8s %10.2e %10.2e %10.2e %10.2e  %s" % r)
ok = [r for r in rows if r[2] is not None]
print("\nsummary over %d runs:" % len(ok))
for name, _, _ in PATHS:
    rr = [r for r in ok if r[1] == name]
    if not rr:
        continue
    print("  %-8s worst res_z %.2e  res_inv %.2e  true|dz| %.2e  true|dss| %.2e ; max true|dz| / res_z = %.1f ; max true|dss| / res_inv = %.1f" % (
        name, max(r[2] for r in rr), max(r[3] for r in rr), max(r[4] for r in rr), max(r[5] for r in rr),
        max(r[4] / max(r[2], 1e-18) for r in rr), max(r[5] / max(r[3], 1e-18) for r in rr)))
for tz, ti in ((1e-10, 1e-9), (2e-10, 2e-9), (1e-9, 1e-8)):
    bad_pass = [r for r in ok if r[2] <= tz and r[3] <= ti and (r[4] > 1e-8 or r[5] > 1e-6)]
    rejected = [r for r in ok if not (r[2] <= tz and r[3] <= ti)]
    needless = [r for r in rejected if r[4] <= 1e-9 and r[5] <= 1e-7]
    print("  tol (%g, %g): passes the probe but misses the bar: %d ; rejected: %d (of which comfortably inside the bar: %d)" % (tz, ti, len(bad_pass), len(rejected), len(needless)))
