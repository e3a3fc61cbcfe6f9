"""Accuracy of the unpivoted shifted sweep against the pivoted path over the randomized test's configurations
(dense cases only): max |dz|, |dss| against the oracle per factor path, grouped by variogram model."""
import sys, collections
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import importlib.util
spec = importlib.util.spec_from_file_location("t", "tests/test_randomized_parity.py")
m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
import pykrige_amd as pa
from oracle import kriging_oracle as ko

worst = collections.defaultdict(lambda: [0.0, 0.0, 0.0, 0.0, 0, 0.0, 0.0])
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 400):
    c = m._case(seed)
    if c["window"] or c["drift"].get("specified") or c["drift"].get("functional"):
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
    zr, sr = ko.execute(st, c["style"], *c["axes"], mask=c["mask"])
    keep = np.ones(c["shape"], bool) if c["mask"] is None else ~c["mask"]
    key = (c["model"], "UK" if c["universal"] else "OK")
    for fac, off in ((1, 0), (2, 2), (1, 5)):  # sweep, pivoted, half sweep (upper block triangle only: option symsweep)
        mdl._get_handle().set_option("factor", fac)
        mdl._get_handle().set_option("symsweep", 1 if off == 5 else 0)
        try:
            z, ss = mdl.execute(c["style"], *c["axes"], mask=c["mask"], backend="loop") if c["mask"] is not None else mdl.execute(c["style"], *c["axes"], backend="loop")
        except Exception:
            worst[key][4] += 1
            continue
        worst[key][off] = max(worst[key][off], float(np.abs(np.ma.getdata(z) - np.ma.getdata(zr))[keep].max()))
        worst[key][off + 1] = max(worst[key][off + 1], float(np.abs(np.ma.getdata(ss) - np.ma.getdata(sr))[keep].max()))
for key in sorted(worst):
    w = worst[key]
    print("%-12s %s  sweep |dz| %.1e |dss| %.1e   pivoted |dz| %.1e |dss| %.1e   half sweep |dz| %.1e |dss| %.1e   sweep refused %d" % (key[0], key[1], w[0], w[1], w[2], w[3], w[5], w[6], w[4]))
