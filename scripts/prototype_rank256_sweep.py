#!/usr/bin/env python
"""CPU prototype (NumPy) of a TWO-PIVOT block Gauss-Jordan sweep -- the rank-256 form of K2's trailing update (the open
item of the inverse at N >= 7000, where the update's MFMA phase and its read-modify-write phase of T add: one pass over T per TWO
128-wide pivots halves the second).  What it pins down before any device code exists:

  * the algebra of the fused step, special tiles included (the rows / columns of the first pivot take their panel-1 write-back value
    as the base of the second update instead of T - C1 R1), against the plain one-pivot sweep the library runs (k_update);
  * the same in the SYMMETRIC half-sweep form the library uses for the SPD-shifted matrix (only the upper block triangle is
    maintained; T_ab = -T_ba^T when exactly one of the blocks a, b has been swept);
  * what a pair of steps reads and writes of T in each form (the traffic model behind the estimate in DESIGN).

    python scripts/prototype_rank256_sweep.py [N] [block]
"""
import sys
import time

import numpy as np


def blocks(n, b):
    return [(i, min(i + b, n)) for i in range(0, n, b)]


def sweep_one_pivot(a, b):
    """Plain in-place block Gauss-Jordan (k_diag_inv + k_panel + k_update): returns the inverse."""
    t = a.copy()
    for (k0, k1) in blocks(t.shape[0], b):
        dinv = np.linalg.inv(t[k0:k1, k0:k1])
        cold = t[:, k0:k1].copy()
        r = dinv @ t[k0:k1, :]
        t -= cold @ r                      # every tile outside row / column k gets its rank-b update (the others are overwritten below)
        t[k0:k1, :] = r
        t[:, k0:k1] = -cold @ dinv
        t[k0:k1, k0:k1] = dinv
    return t


def sweep_two_pivots(a, b, count=None):
    """Two pivots per pass over T.  Per pair (k, k+1):
         1. panel 1 from T as it stands: Dinv1, Cold1 = T[:, k], R1 = Dinv1 T[k, :], Cnew1 = -Cold1 Dinv1;
         2. COLUMN PART (rank b, block column and block row k+1 only): T[i, k+1] -= Cold1[i] R1[k+1], T[k+1, j] -= Cold1[k+1] R1[j];
            the two tiles that touch pivot 1 take their write-back values: T[k, k+1] = R1[k+1], T[k+1, k] = Cnew1[k+1];
         3. panel 2 from that column / row: Dinv2, Cold2 = T[:, k+1], R2 = Dinv2 T[k+1, :], Cnew2 = -Cold2 Dinv2;
         4. FUSED REST, one read-modify-write of every remaining tile:
              i, j not in {k, k+1}:   T[i, j] -= Cold1[i] R1[j] + Cold2[i] R2[j]                 (rank 2b)
              row k:                   T[k, j]  = R1[j]    - Cold2[k] R2[j]                        (base = panel-1 write-back)
              column k:                T[i, k]  = Cnew1[i] - Cold2[i] R2[k]
              (k, k):                  T[k, k]  = Dinv1    - Cold2[k] R2[k]
              row / column k+1:        T[k+1, j] = R2[j],  T[i, k+1] = Cnew2[i],  T[k+1, k+1] = Dinv2."""
    t = a.copy()
    n = t.shape[0]
    bl = blocks(n, b)
    p = 0
    while p < len(bl):
        k0, k1 = bl[p]
        if p + 1 == len(bl):  # an odd block column at the end: one plain step
            dinv = np.linalg.inv(t[k0:k1, k0:k1])
            cold = t[:, k0:k1].copy()
            r = dinv @ t[k0:k1, :]
            t -= cold @ r
            t[k0:k1, :] = r
            t[:, k0:k1] = -cold @ dinv
            t[k0:k1, k0:k1] = dinv
            if count is not None:
                count["rmw_tiles"] += (len(bl) - 1) ** 2
            break
        m0, m1 = bl[p + 1]
        # 1. panel 1
        dinv1 = np.linalg.inv(t[k0:k1, k0:k1])
        cold1 = t[:, k0:k1].copy()
        r1 = dinv1 @ t[k0:k1, :]
        cnew1 = -cold1 @ dinv1
        # 2. column part: block column / row k+1
        t[:, m0:m1] -= cold1 @ r1[:, m0:m1]
        t[m0:m1, :] -= cold1[m0:m1] @ r1
        t[m0:m1, m0:m1] += cold1[m0:m1] @ r1[:, m0:m1]       # (the corner was hit twice)
        t[k0:k1, m0:m1] = r1[:, m0:m1]
        t[m0:m1, k0:k1] = cnew1[m0:m1]
        # 3. panel 2
        dinv2 = np.linalg.inv(t[m0:m1, m0:m1])
        cold2 = t[:, m0:m1].copy()
        r2 = dinv2 @ t[m0:m1, :]
        cnew2 = -cold2 @ dinv2
        # 4. fused rest
        rest = np.ones(n, bool)
        rest[k0:k1] = rest[m0:m1] = False
        ri = np.flatnonzero(rest)
        t[np.ix_(ri, ri)] -= cold1[ri] @ r1[:, ri] + cold2[ri] @ r2[:, ri]
        t[k0:k1, ri] = r1[:, ri] - cold2[k0:k1] @ r2[:, ri]
        t[ri, k0:k1] = cnew1[ri] - cold2[ri] @ r2[:, k0:k1]
        t[k0:k1, k0:k1] = dinv1 - cold2[k0:k1] @ r2[:, k0:k1]
        t[m0:m1, :] = r2
        t[:, m0:m1] = cnew2
        t[m0:m1, m0:m1] = dinv2
        if count is not None:
            nb = len(bl)
            count["rmw_tiles"] += (nb - 2) ** 2        # fused rest: every tile outside the two pivots' rows / columns, once
            count["rmw_tiles_column_part"] += 2 * (nb - 1)
        p += 2
    return t


def half_sweep_two_pivots(a, b):
    """The same pair step on the UPPER block triangle only, for a symmetric matrix swept without pivoting (the library's half sweep):
    tile (i, j), i <= j, is kept; the lower one follows from T_ji = s_i s_j T_ij^T with s = -1 for swept blocks, +1 otherwise (after
    the whole sweep every block is swept and the inverse is symmetric again).  The prototype keeps a full copy for the comparison and
    reads ONLY upper tiles when it forms panels and updates -- the access pattern a device kernel would have."""
    n = a.shape[0]
    bl = blocks(n, b)
    nb = len(bl)
    up = np.triu(np.ones((nb, nb), bool))
    t = a.copy()
    swept = np.zeros(nb, bool)

    def tile(i, j):  # T_ij from the upper triangle alone
        if i <= j:
            return t[bl[i][0]:bl[i][1], bl[j][0]:bl[j][1]]
        s = (-1.0 if swept[i] else 1.0) * (-1.0 if swept[j] else 1.0)
        return s * t[bl[j][0]:bl[j][1], bl[i][0]:bl[i][1]].T

    def column(k):
        return np.vstack([tile(i, k) for i in range(nb)])

    def row(k):
        return np.hstack([tile(k, j) for j in range(nb)])

    p = 0
    while p < nb:
        pair = [p] if p + 1 == nb else [p, p + 1]
        cold, r, cnew, dinv = {}, {}, {}, {}
        for q, k in enumerate(pair):
            k0, k1 = bl[k]
            if q == 1:  # column part of pivot 1 on block column / row k (upper tiles only), then pivot 1's write-back tiles touching k
                kp = pair[0]
                for i in range(nb):
                    if i == kp:
                        continue
                    if i <= k:
                        t[bl[i][0]:bl[i][1], k0:k1] -= cold[kp][bl[i][0]:bl[i][1]] @ r[kp][:, k0:k1]
                    else:
                        t[k0:k1, bl[i][0]:bl[i][1]] -= cold[kp][k0:k1] @ r[kp][:, bl[i][0]:bl[i][1]]
                t[bl[kp][0]:bl[kp][1], k0:k1] = r[kp][:, k0:k1]  # tile (kp, k), kp < k: upper
            dinv[k] = np.linalg.inv(tile(k, k))
            cold[k] = column(k)
            r[k] = dinv[k] @ row(k)
            cnew[k] = -cold[k] @ dinv[k]
            if q == 0 and len(pair) == 2:
                swept[k] = True  # panel 2 reads tile (k+1, k) as the mirror of (k, k+1): pivot 1 counts as swept from here on
        # fused rest on the upper tiles
        for i in range(nb):
            for j in range(i, nb):
                i0, i1 = bl[i]
                j0, j1 = bl[j]
                if len(pair) == 2:
                    ka, kb_ = pair
                    if i == kb_ or j == kb_:
                        continue  # written back below
                    if i == ka or j == ka:
                        base = dinv[ka] if (i == ka and j == ka) else (r[ka][:, j0:j1] if i == ka else cnew[ka][i0:i1])
                        t[i0:i1, j0:j1] = base - cold[kb_][i0:i1] @ r[kb_][:, j0:j1]
                    else:
                        t[i0:i1, j0:j1] -= cold[ka][i0:i1] @ r[ka][:, j0:j1] + cold[kb_][i0:i1] @ r[kb_][:, j0:j1]
                else:
                    ka = pair[0]
                    if i == ka or j == ka:
                        continue
                    t[i0:i1, j0:j1] -= cold[ka][i0:i1] @ r[ka][:, j0:j1]
        kl = pair[-1]
        k0, k1 = bl[kl]
        for j in range(kl, nb):
            t[k0:k1, bl[j][0]:bl[j][1]] = r[kl][:, bl[j][0]:bl[j][1]]
        for i in range(kl):
            t[bl[i][0]:bl[i][1], k0:k1] = cnew[kl][bl[i][0]:bl[i][1]]
        t[k0:k1, k0:k1] = dinv[kl]
        for k in pair:
            swept[k] = True
        p += len(pair)
    # mirror: every block swept -> T_ji = T_ij^T
    out = np.triu(t)
    for i in range(nb):
        for j in range(i + 1, nb):
            out[bl[j][0]:bl[j][1], bl[i][0]:bl[i][1]] = t[bl[i][0]:bl[i][1], bl[j][0]:bl[j][1]].T
    for i in range(nb):
        out[bl[i][0]:bl[i][1], bl[i][0]:bl[i][1]] = t[bl[i][0]:bl[i][1], bl[i][0]:bl[i][1]]
    return out


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1100
    b = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    rng = np.random.default_rng(4)
    x = rng.random((n - 1, 2))
    d = np.hypot(x[:, None, 0] - x[None, :, 0], x[:, None, 1] - x[None, :, 1])
    gam = 1.0 - np.exp(-d / 0.1) + 0.01 * (d > 0)
    s = 1.01
    # the library's SPD-shifted kriging matrix: C = s 11^T - Gamma bordered by the unbiasedness row (symmetric, swept without pivoting)
    a = np.zeros((n, n))
    a[:n - 1, :n - 1] = s - gam
    a[n - 1, :n - 1] = a[:n - 1, n - 1] = 1.0
    a[n - 1, n - 1] = 0.0
    a[:n - 1, :n - 1] += 0.0
    ref = np.linalg.inv(a)
    t0 = time.time()
    one = sweep_one_pivot(a, b)
    cnt = {"rmw_tiles": 0, "rmw_tiles_column_part": 0}
    two = sweep_two_pivots(a, b, cnt)
    half = half_sweep_two_pivots(a, b)
    sc = np.abs(ref).max()
    print("N = %d, block %d (%d block columns), cond_1 %.1e" % (n, b, len(blocks(n, b)), np.linalg.cond(a, 1)))
    print("one pivot per pass   vs LAPACK: max|diff| / max|inv| %.2e" % (np.abs(one - ref).max() / sc))
    print("two pivots per pass  vs LAPACK: max|diff| / max|inv| %.2e   vs one pivot per pass: %.2e" % (
        np.abs(two - ref).max() / sc, np.abs(two - one).max() / sc))
    print("two pivots, upper block triangle only (half sweep) vs LAPACK: %.2e   vs one pivot per pass: %.2e" % (
        np.abs(half - ref).max() / sc, np.abs(half - one).max() / sc))
    nb = len(blocks(n, b))
    print("read-modify-write passes over tiles of T, full sweep: one pivot per pass %d, two pivots per pass %d (+ %d in the column parts)" % (
        nb * (nb - 1) ** 2, cnt["rmw_tiles"], cnt["rmw_tiles_column_part"]))
    for big in (5000, 8000):
        nbb = (big + 1 + 127) // 128
        up = nbb * (nbb + 1) // 2
        print("N = %d (%d block columns), half sweep: upper tiles %d; per PAIR of pivots  one-pivot form 2 x %d RMW tiles x (128 KB read + 128 KB "
              "write) = %.0f MB,  two-pivot form %d + %d = %.0f MB;  MFMA work the same (2 x 128^3 x 2 flop per tile and pivot)" % (
                  big, nbb, up, up, 2 * up * 0.262144, up, 2 * nbb, (up + 2 * nbb) * 0.262144))
    print("(%.1f s)" % (time.time() - t0))
