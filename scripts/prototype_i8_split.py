#!/usr/bin/env python
"""Gate 1 of the round-4 review's feasibility study: an fp64-equivalent contraction on the int8 matrix pipe by an error-free split
(Ozaki scheme: Ootomo et al., "DGEMM on integer matrix multiplication unit", 2024).  CPU / NumPy emulation on the full-size reference
slabs tests/golden/fullsize/c2.npz (OK2D N = 5000, exponential, cond_1 4.6e6) and c4.npz (UK N = 4000 + drift).

What K3b computes per point t:  W = A_inv b_t  (the dense product, 2 M^2 flops),  sigma^2_t = -b_t . W,  z_t = c . b_t.
The split: every ROW of A_inv is scaled by a power of two to (-1, 1) and cut into s signed 7-bit slices (int8), every COLUMN b_t
likewise; slice products accumulate EXACTLY in int32 (K <= 2^17), the s (s + 1) / 2 products with p + q <= s + 1 are recombined in
fp64 with their weights 2^(-7 (p + q)).  Each slice product is one int8 GEMM of the full size.

Needed for the gate: max|dz| <= 1e-10 and max|dsigma^2| <= 1e-9 against the reference with s <= 7 (28 GEMMs: at 3.9 POPS int8 that is
~ 140 TFLOP/s-equivalent, 1.8 x the fp64 matrix pipe); otherwise stop.

    python scripts/prototype_i8_split.py [c2|c4] [npoints]
"""
import os
import sys
import time

import numpy as np
import scipy.linalg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import kriging_oracle as ko  # noqa: E402
from tests import _fixtures as fx  # noqa: E402


def split(x, s, axis):
    """x (fp64) -> (slices[s] of integer-valued fp64 in [-127, 127], scale): x ~= scale * sum_p slices[p] 2^(-7 (p + 1)), scale a power
    of two per row (axis = 1) / per column (axis = 0) with |x / scale| < 1.  Truncation towards zero, as a shift would do."""
    amax = np.abs(x).max(axis=axis, keepdims=True)
    e = np.ceil(np.log2(np.where(amax > 0, amax, 1.0)))
    e = np.where(amax >= 2.0 ** e, e + 1, e)  # strict |x| < 2^e
    scale = 2.0 ** e
    r = x / scale
    out = []
    for _ in range(s):
        r = r * 128.0
        q = np.trunc(r)
        out.append(q)
        r = r - q
    return out, scale


def contraction_split(ainv, bt, s):
    """W = ainv @ bt by the split; exact integer slice products (float64 holds them exactly: |sum| <= K 127^2 < 2^53)."""
    sa, ra = split(ainv, s, 1)   # rows of A_inv
    sb, cb = split(bt, s, 0)     # columns of B (one per point)
    w = np.zeros((ainv.shape[0], bt.shape[1]))
    n_gemm = 0
    # smallest weights first: the recombination's own rounding then stays below the truncation error
    for tot in range(s + 1, 1, -1):
        acc = np.zeros_like(w)
        for p in range(1, s + 1):
            q = tot - p
            if 1 <= q <= s:
                acc += sa[p - 1] @ sb[q - 1]
                n_gemm += 1
        w += acc * 2.0 ** (-7 * tot)
    return w * ra * cb, n_gemm


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "c2"
    npt = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    with np.load(os.path.join(fx.GOLDEN, "fullsize", name + ".npz"), allow_pickle=False) as f:
        g = {k: f[k] for k in f.files}
    st = fx.state_from(name, g)
    a = ko.kriging_matrix(st)
    t0 = time.time()
    ainv = scipy.linalg.inv(a)
    cond1 = np.abs(a).sum(0).max() * np.abs(ainv).sum(0).max()
    axes = fx.grid_args(g)
    X, Y = np.meshgrid(*axes)
    sel = np.unique(np.concatenate([np.arange(8), np.linspace(0, X.size - 1, npt).astype(int)]))
    # the 8 exact-hit nodes of the slab as well
    shape = (len(g["gridx"]), len(g["gridy"]))
    idx = np.unravel_index(g["node_flat"], shape)
    sel = np.unique(np.concatenate([sel, idx[1] * shape[0] + idx[0]]))
    pts = np.stack([X.ravel(), Y.ravel()], 1)[sel]
    pts_adj = ko.adjust_for_anisotropy(pts, st.center, st.scaling, st.angle)
    b = ko.rhs(st, pts_adj)          # (npt, M)
    n = st.n
    zref, sref = g["z"].ravel()[sel], g["ss"].ravel()[sel]
    w64 = ainv @ b.T
    z64 = (w64[:n] * st.values[:, None]).sum(0)
    s64 = -(w64 * b.T).sum(0)
    print("%s: N = %d, M = %d, %d points, cond_1 %.2e (%.0f s)" % (name, n, a.shape[0], sel.size, cond1, time.time() - t0))
    print("  fp64 product              : max|dz| %.2e  max|dss| %.2e   (against the stored reference slab)" % (np.abs(z64 - zref).max(), np.abs(s64 - sref).max()))
    # z = c . b needs no big product (c = A_inv[:, :N] Z is a vector); the split matters for sigma^2 only -- both are reported
    for s in range(4, 11):
        t1 = time.time()
        w, ng = contraction_split(ainv, b.T, s)
        zs = (w[:n] * st.values[:, None]).sum(0)
        ss = -(w * b.T).sum(0)
        print("  s = %2d slices, %2d int8 GEMMs: max|dz| %.2e  max|dss| %.2e   vs the fp64 product: |dz| %.2e |dss| %.2e   (%.0f s)" % (
            s, ng, np.abs(zs - zref).max(), np.abs(ss - sref).max(), np.abs(zs - z64).max(), np.abs(ss - s64).max(), time.time() - t1))
    # the same with the symmetric two-sided scaling D A_inv D, D = diag(1 / sqrt|a_ii|) folded into the split's row scales: does equilibration help?
    d = 1.0 / np.sqrt(np.abs(np.diag(ainv)))
    ainv_e = ainv * d[:, None] * d[None, :]
    be = b.T / d[:, None]
    for s in (6, 7, 8):
        w, ng = contraction_split(ainv_e, be, s)
        w = w / d[:, None]
        ss = -(w * b.T).sum(0)
        print("  equilibrated, s = %d: max|dss| %.2e vs reference, %.2e vs the fp64 product" % (s, np.abs(ss - sref).max(), np.abs(ss - s64).max()))


if __name__ == "__main__":
    main()
