"""Round 6, gate 1 of the review's item 3 (CPU, NumPy): K2 + K3 in FACTOR form -- Cholesky + triangular inverse instead of the Gauss-Jordan sweep.

With s = psill + nugget, C = s 11^T - Gamma (the covariance matrix: SPD for the bounded models) = L L^T, W = L^-1, w = C^-1 1, mu = 1^T w, and for
a point c = s - gamma(d) (the covariance vector; b = [c - s 1; 1] in the reference's system, ok.py:669-676):
    sigma^2 = s - |W c|^2 + (1 - w.c)^2 / mu          z = cvec . b  (unchanged)
K2 becomes potrf + trtri (2/3 M^3 flops instead of the half sweep's M^3), K3 a triangular product W c (M^2 flops per point, as the symmetric half product) whose
epilogue is the square of its own accumulators.  This script measures what the form does to PARITY on the stored reference slabs (tests/golden/fullsize c2, c5: OK) and c4
(UK: drift columns F by the Schur complement F^T C^-1 F), next to today's form (-b^T A^-1 b with A^-1 from LAPACK), and counts flops and bytes per step.
    python scripts/prototype_cholesky_form.py [c2 c4 c5]
"""
import os
import sys
import time

import numpy as np
import scipy.linalg
from scipy.spatial.distance import cdist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import kriging_oracle as ko  # noqa: E402  (the checker's variogram functions; this script is a prototype, not product)


def run(name):
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", "fullsize", name + ".npz"), allow_pickle=False))
    model = str(g["model"])
    par = ko.internal_parameters(model, g["params_user"].tolist())
    x, y, v = g["x"], g["y"], g["v"]
    n = x.size
    X, Y = np.meshgrid(g["gridx"], g["gridy"])
    pts = np.stack([X.ravel(), Y.ravel()], 1)
    sta = np.stack([x, y], 1)
    vf = lambda m_, d_: ko.variogram(model, m_, d_)  # noqa: E731
    s = par[0] + par[2]
    t0 = time.time()
    gam = vf(par, cdist(sta, sta))
    np.fill_diagonal(gam, 0.0)
    C = s - gam  # diagonal: s (gamma(0) = 0 by the reference's fill_diagonal, ok.py:644)
    L = np.linalg.cholesky(C)
    W = scipy.linalg.solve_triangular(L, np.eye(n), lower=True)
    # drift columns (UK c4: regional linear x, y + point_log wells), equilibrated like the library does; OK: just the ones column
    F = [np.ones(n)]
    fp = [np.ones(pts.shape[0])]
    if "wells" in g:
        F += [x, y]
        fp += [pts[:, 0], pts[:, 1]]
        for wx, wy, ws in g["wells"]:
            F.append(ko._log_well(np.hypot(x - wx, y - wy), ws))
            fp.append(ko._log_well(np.hypot(pts[:, 0] - wx, pts[:, 1] - wy), ws))
    F, fp = np.stack(F, 1), np.stack(fp, 1)
    sc = np.ones(F.shape[1])
    for j in range(1, F.shape[1]):  # equilibration: (f - mean) / max|f - mean| (same span with the ones column)
        cj = F[:, j].mean()
        sj = 1.0 / np.abs(F[:, j] - cj).max()
        F[:, j] = sj * (F[:, j] - cj)
        fp[:, j] = sj * (fp[:, j] - cj)
    WF = W @ F                       # L^-1 F
    S = WF.T @ WF                    # F^T C^-1 F  (p+1 x p+1, SPD)
    Ls = np.linalg.cholesky(S)
    t_fac = time.time() - t0
    # per point
    cpt = s - vf(par, cdist(pts, sta))
    hit = cdist(pts, sta) <= 1e-10
    cpt[hit] = s                     # the eps rule: gamma := 0 at an exact hit (ok.py:672-676), i.e. covariance s
    Yv = cpt @ W.T                   # rows y = W c
    r = fp - Yv @ WF                 # f0 - F^T C^-1 c
    q = scipy.linalg.solve_triangular(Ls, r.T, lower=True).T
    ss_new = s - (Yv * Yv).sum(1) + (q * q).sum(1)
    # z: weights lambda = C^-1 (c + F nu), nu = S^-1 r  ->  z = lambda . v
    nu = scipy.linalg.solve_triangular(Ls.T, q.T, lower=False).T
    alpha = W.T @ (W @ v)            # C^-1 v
    beta = WF.T @ (W @ v)            # F^T C^-1 v
    z_new = cpt @ alpha + nu @ beta
    zr, sr = g["z"].ravel(), g["ss"].ravel()
    # today's form with LAPACK's inverse (what the library's sweep is probed against)
    m = n + F.shape[1]
    A = np.zeros((m, m))
    A[:n, :n] = -gam
    A[:n, n:] = F
    A[n:, :n] = F.T
    Ainv = scipy.linalg.inv(A)
    b = np.concatenate([-(s - cpt), fp], 1)
    ss_old = -np.einsum("ij,ij->i", b @ Ainv, b)
    z_old = (b @ Ainv)[:, :n] @ v
    print("%s: N = %d, %d points, cond_1 %.2e; cond_2(C) %.2e" % (name, n, pts.shape[0], float(g["cond1"]), np.linalg.cond(C)))
    print("   factor form (Cholesky + triangular inverse):  max|dz| %.3e   max|dsigma^2| %.3e   (min sigma^2 %.3e at exact hits: %.3e)" % (
        np.abs(z_new - zr).max(), np.abs(ss_new - sr).max(), ss_new.min(), np.abs(ss_new[hit.any(1)]).max() if hit.any() else float("nan")))
    print("   today's form  (-b^T A^-1 b, LAPACK inverse):  max|dz| %.3e   max|dsigma^2| %.3e" % (np.abs(z_old - zr).max(), np.abs(ss_old - sr).max()))
    M = m
    print("   flops per factorisation: half sweep M^3 = %.3g, potrf + trtri 2/3 M^3 = %.3g;  per point: both M^2 = %.3g" % (M ** 3, 2 / 3 * M ** 3, M ** 2))
    nb = (M + 127) // 128
    rmw_sweep = nb * (nb * (nb + 1) / 2) * 128 * 128 * 8 * 2
    rmw_chol = 2 * sum((nb - k) * (nb - k + 1) / 2 for k in range(nb)) * 128 * 128 * 8 * 2
    print("   read-modify-write traffic of the trailing updates (128-blocks): half sweep %.1f GB, potrf + trtri %.1f GB   (CPU factor %.1f s)" % (
        rmw_sweep / 1e9, rmw_chol / 1e9, t_fac))


if __name__ == "__main__":
    for nm in (sys.argv[1:] or ["c2", "c4", "c5"]):
        run(nm)
