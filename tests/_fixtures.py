"""Helpers shared by the oracle tests (CPU) and the HIP parity tests (GPU)."""
import glob
import os

import numpy as np

from oracle import kriging_oracle as ko

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# functional drifts cannot be stored in an .npz; they are re-declared here exactly as in oracle/make_golden.py
FUNCS = {
    "uk2d_spec_func": [lambda a, b: a * b, lambda a, b: a**2],
    "uk3d_rl_func": [lambda a, b, c: a * c],
}


def names():
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN, "*.npz")))


def load(name):
    with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as f:
        return {k: f[k] for k in f.files}


def state_from(name, g):
    """Oracle KrigingState for a golden fixture."""
    ndim = 3 if "zc" in g else 2
    coords = np.stack([g["x"], g["y"]] + ([g["zc"]] if ndim == 3 else []), axis=1)
    model = str(g["model"])
    scaling = np.atleast_1d(g["scaling"]).tolist() if "scaling" in g else [1.0] * (ndim - 1)
    angle = np.atleast_1d(g["angle"]).tolist() if "angle" in g else [0.0] * (2 * ndim - 3)
    rl = bool(g["regional_linear"]) if "regional_linear" in g else name in ("uk2d_spec_func", "uk3d_rl_func")
    return ko.KrigingState(
        ndim=ndim, coords_orig=coords, values=g["v"], model=model,
        params=ko.internal_parameters(model, g["params_user"].tolist()),
        scaling=scaling, angle=angle,
        exact_values=bool(g["exact"]) if "exact" in g else True,
        regional_linear=rl,
        point_log=g["wells"] if "wells" in g else None,
        specified_data=[g["spec_data"]] if "spec_data" in g else [],
        functional=FUNCS.get(name, []),
    )


def grid_args(g):
    if "gridz" in g:
        return (g["gridx"], g["gridy"], g["gridz"])
    return (g["gridx"], g["gridy"])
