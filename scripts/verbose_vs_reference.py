"""Differential run (GPU box): what the real reference (oracle/_ref) and the drop-in PRINT with verbose=True -- constructors of the four classes (fitted and given
variograms, every model family, drifts, statistics), update_variogram_model, print_statistics, execute -- compared line by line (numbers to 6 significant
digits: the statistics come from different arithmetic).  `--cpu`: only the cases that need no device (OrdinaryKriging without statistics).  Exits non-zero on a difference."""
import contextlib
import difflib
import io
import re
import sys

import numpy as np

sys.path.insert(0, ".")
import pykrige_amd as pa  # noqa: E402
from oracle import ref_package as rp  # noqa: E402

pk = rp.import_reference(stub_statistics=False)
cpu = "--cpu" in sys.argv
rng = np.random.default_rng(3)
n = 25
x, y, z3, v = rng.random(n), rng.random(n), rng.random(n), rng.random(n)
gx = np.linspace(0, 1, 4)
ext = dict(external_drift=np.arange(20.0).reshape(4, 5), external_drift_x=np.linspace(-0.1, 1.1, 5), external_drift_y=np.linspace(-0.1, 1.1, 4))
cases = {
    "OK fitted spherical": (lambda m: m.ok.OrdinaryKriging(x, y, v, variogram_model="spherical", verbose=True), True),
    "OK given linear, then updated": (lambda m: m.ok.OrdinaryKriging(x, y, v, variogram_model="linear", variogram_parameters=[1.0, 0.1], verbose=True), True),
    "OK power fitted": (lambda m: m.ok.OrdinaryKriging(x, y, v, variogram_model="power", verbose=True), True),
    "OK custom": (lambda m: m.ok.OrdinaryKriging(x, y, v, variogram_model="custom", variogram_parameters=[1.0, 0.2], variogram_function=lambda p, d: p[0] * d + p[1], verbose=True), True),
    "OK geographic + anisotropy": (lambda m: m.ok.OrdinaryKriging(x * 90, y * 40, v, variogram_model="exponential", coordinates_type="geographic", anisotropy_scaling=2.0, verbose=True), True),
    "OK plotting flag": (lambda m: m.ok.OrdinaryKriging(x, y, v, variogram_model="gaussian", variogram_parameters=[1.0, 0.3, 0.1], verbose=True, enable_plotting=False), True),
    "OK statistics": (lambda m: m.ok.OrdinaryKriging(x, y, v, variogram_model="hole-effect", verbose=True, enable_statistics=True), False),
    "UK regional + wells": (lambda m: m.uk.UniversalKriging(x, y, v, variogram_model="gaussian", drift_terms=["regional_linear", "point_log"], point_drift=[[0.1, 0.2, 1.0]], verbose=True), False),
    "UK external + specified + functional": (lambda m: m.uk.UniversalKriging(x, y, v, variogram_model="hole-effect", drift_terms=["external_Z", "specified", "functional"], specified_drift=[x],
                                                                             functional_drift=[lambda a, b: a], verbose=True, **ext), False),
    "OK3D": (lambda m: m.ok3d.OrdinaryKriging3D(x, y, z3, v, variogram_model="linear", verbose=True), False),
    "UK3D": (lambda m: m.uk3d.UniversalKriging3D(x, y, z3, v, variogram_model="power", drift_terms=["regional_linear", "specified", "functional"], specified_drift=[x],
                                                 functional_drift=[lambda a, b, c: a], verbose=True), False),
}


def norm(s):
    return re.sub(r"-?\d+\.\d+(e[-+]?\d+)?", lambda m: "%.6g" % float(m.group(0)), s)


bad = 0
for name, (mk, on_cpu) in cases.items():
    if cpu and not on_cpu:
        continue
    outs = []
    for mod in (pk, pa):
        buf = io.StringIO()
        try:
            with contextlib.redirect_stdout(buf):
                o = mk(mod)
                if not cpu:
                    if o._ndim == 3 if hasattr(o, "_ndim") else hasattr(o, "Z_ORIG"):
                        o.update_variogram_model("exponential", [1.0, 0.3, 0.05], anisotropy_scaling_y=2.0)
                        o.execute("grid", gx, gx, gx)
                    else:
                        o.update_variogram_model("exponential", [1.0, 0.3, 0.05], anisotropy_scaling=2.0)
                        if "specified" not in name:
                            o.execute("grid", gx, gx)
                    o.print_statistics()
                    o.switch_verbose()
                    o.update_variogram_model("gaussian", [1.0, 0.3, 0.05])
        except Exception as e:  # noqa: BLE001
            buf.write("RAISED %s %s" % (type(e).__name__, e))
        outs.append(norm(buf.getvalue()))
    same = outs[0] == outs[1]
    print("%-40s %s (%d lines)" % (name, "same" if same else "DIFFERENT", len(outs[0].splitlines())))
    if not same:
        bad += 1
        for ln in difflib.unified_diff(outs[0].splitlines(), outs[1].splitlines(), "reference", "drop-in", lineterm="", n=0):
            print("    " + ln[:170])
print("%d case(s) differ" % bad)
sys.exit(1 if bad else 0)
