"""CPU prototype (NumPy; no GPU, not product code) of a BLOCK one-sided Jacobi pseudo-inverse for the symmetric kriging matrix --
groundwork for replacing the scalar one-sided Jacobi fallback of `run_pseudo_inverse` (factor_path 4: M - 1 launches per sweep, every
launch streams the whole matrix: 9.3 s at M = 4000).  Blocks of b rows; a step takes a pair of row blocks X (2b x M), forms the Gram
matrix G = X X^T (one pass over the rows), diagonalises it (2b x 2b, would live in LDS) and replaces X by Q^T X, W likewise (second
pass): per sweep the matrix is streamed ~3 (M / b - 1) times instead of ~2 (M - 1) times.

Question this script answers: does going through Gram matrices keep the accuracy the pseudo-inverse needs (singular values down to the
cut-off M eps sigma_max must come out with small RELATIVE error, since 1 / sigma is what the result is made of)?  Compared against
scipy.linalg.pinv on the test suite's hard inputs.   usage: python scripts/prototype_block_jacobi.py [b] [inner]   inner = eigh | jacobi"""
import sys, time
import numpy as np
import scipy.linalg

sys.path.insert(0, ".")
from oracle import kriging_oracle as ko
from tests import _fixtures as fx

EPS = np.finfo(float).eps


def inner_jacobi(G, sweeps=12):
    """Two-sided cyclic Jacobi on the symmetric positive semi-definite G (round-robin ordering, rotations of one round
    applied together): eigenvectors Q with G ~ Q diag Q^T.  On graded matrices Jacobi keeps relative accuracy (Demmel-Veselic)."""
    n0 = G.shape[0]
    if n0 % 2:  # the round-robin needs an even order: one idle slot
        Gp = np.zeros((n0 + 1, n0 + 1))
        Gp[:n0, :n0] = G
        return inner_jacobi(Gp, sweeps)[:n0, :n0]
    n = n0
    A = G.copy()
    Q = np.eye(n)
    idx = list(range(n))
    for _ in range(sweeps):
        off = 0.0
        for _r in range(n - 1):
            p = np.array(idx[: n // 2])
            q = np.array(idx[n // 2:][::-1])
            lo, hi = np.minimum(p, q), np.maximum(p, q)
            app, aqq, apq = A[lo, lo], A[hi, hi], A[lo, hi]
            den = np.sqrt(np.abs(app * aqq))
            rel = np.where(den > 0, np.abs(apq) / np.where(den > 0, den, 1.0), 0.0)
            off = max(off, float(rel.max()))
            act = rel > 1e-16
            if act.any():
                with np.errstate(divide="ignore", invalid="ignore"):
                    zeta = np.where(act, (aqq - app) / (2.0 * np.where(act, apq, 1.0)), 0.0)
                t = np.where(act, np.sign(zeta + (zeta == 0)) / (np.abs(zeta) + np.sqrt(1.0 + zeta * zeta)), 0.0)
                c = 1.0 / np.sqrt(1.0 + t * t)
                s = c * t
                J = np.eye(n)
                J[lo, lo], J[hi, hi], J[lo, hi], J[hi, lo] = c, c, s, -s
                A = J.T @ A @ J
                Q = Q @ J
            idx = [idx[0]] + [idx[-1]] + idx[1:-1]
        if off < 1e-15:
            break
    return Q


def block_jacobi_pinv(A, b=32, inner="jacobi", max_sweeps=40, tol=1e-15, sort_rows=True):
    M = A.shape[0]
    B = A.astype(float).copy()
    W = np.eye(M)
    nb = -(-M // b)
    if nb % 2:
        nb += 1
    sweeps = 0
    for sweep in range(max_sweeps):
        sweeps += 1
        n2 = np.einsum("ij,ij->i", B, B)
        smax = float(np.sqrt(n2.max()))
        order = np.argsort(-n2) if sort_rows else np.arange(M)
        blocks = [order[i * b:(i + 1) * b] for i in range(nb)]
        ring = list(range(nb))
        worst = 0.0
        for _r in range(nb - 1):
            for k in range(nb // 2):
                I, J = blocks[ring[k]], blocks[ring[nb - 1 - k]]
                ix = np.concatenate([I, J])
                if ix.size == 0:
                    continue
                X = B[ix]
                G = X @ X.T
                d = np.sqrt(np.abs(np.diag(G)))
                live = d > 0.01 * M * EPS * smax  # rows below a hundredth of the cut-off are the null space: their angles are noise
                if live.sum() > 1:
                    C = G[np.ix_(live, live)] / np.outer(d[live], d[live])
                    worst = max(worst, float(np.abs(C - np.diag(np.diag(C))).max()))
                if inner == "eigh":
                    _, Q = np.linalg.eigh(G)
                else:
                    Q = inner_jacobi(G)
                B[ix] = Q.T @ X
                W[ix] = Q.T @ W[ix]
            ring = [ring[0]] + [ring[-1]] + ring[1:-1]
        if worst < tol:
            break
    sig2 = np.einsum("ij,ij->i", B, B)
    sig = np.sqrt(sig2)
    keep = sig > M * EPS * sig.max()
    P = (B[keep].T / sig2[keep]) @ W[keep]
    return P, sweeps, int((~keep).sum()), worst


def cases():
    (x, y), v = fx.synth(4000 + 301, 301, 2)
    x[-6:], y[-6:] = x[:6], y[:6]
    st = ko.KrigingState(ndim=2, coords_orig=np.stack([x, y], 1), values=v, model="exponential", params=ko.internal_parameters("exponential", [1.0, 0.5, 0.0]))
    yield "OK2D n=301, six duplicated stations, zero nugget", ko.kriging_matrix(st)
    rng = np.random.default_rng(77)
    n = 60
    x = rng.random(n)
    st = ko.KrigingState(ndim=2, coords_orig=np.stack([x, 2.0 * x + 0.25], 1), values=np.sin(4 * x), model="exponential",
                         params=ko.internal_parameters("exponential", [1.0, 0.5, 0.05]), regional_linear=True)
    yield "UK2D n=60, collinear stations + regional-linear drift (rank M - 1)", ko.kriging_matrix(st)
    for tag, nug, dups in (("b", 0.02, 0), ("c", 0.0, 3)):
        n2 = 350
        x2 = rng.random(n2)
        if dups:
            x2[-dups:] = x2[:dups]
        st = ko.KrigingState(ndim=2, coords_orig=np.stack([x2, 0.5 * x2 - 0.1], 1), values=np.cos(3 * x2), model="exponential",
                             params=ko.internal_parameters("exponential", [1.0, 0.5, nug]), regional_linear=True)
        yield "UK2D n=%d collinear + drift, nugget %g, %d duplicates (case %s of the GPU test, smaller)" % (n2, nug, dups, tag), ko.kriging_matrix(st)


if __name__ == "__main__":
    b = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    inner = sys.argv[2] if len(sys.argv) > 2 else "jacobi"
    print("block one-sided Jacobi pseudo-inverse, b = %d, inner solver %s" % (b, inner))
    for name, A in cases():
        ref = scipy.linalg.pinv(A)
        s = np.linalg.svd(A, compute_uv=False)
        t = time.time()
        P, sweeps, ndrop, worst = block_jacobi_pinv(A, b=b, inner=inner)
        rank_ref = int((s > A.shape[0] * EPS * s[0]).sum())
        print("%s: M = %d, rank %d, smallest kept sigma / sigma_max %.1e | %d sweeps, %d rows dropped (scipy drops %d), last max cosine %.1e, "
              "max|P - pinv| / max|pinv| = %.2e   [%.1f s]" % (name, A.shape[0], rank_ref, s[rank_ref - 1] / s[0], sweeps, ndrop, A.shape[0] - rank_ref, worst,
                                                          np.abs(P - ref).max() / np.abs(ref).max(), time.time() - t), flush=True)
