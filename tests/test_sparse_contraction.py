"""Range-aware contraction for the reference's compact-support variogram (spherical: constant beyond the range,
variogram_models.py:56-70): sigma^2 = 2 s - delta^T A_inv delta, z = c . delta with delta = b + s u (include/mikrige.h, option
"sparse", "sparse_rows"; k_rhs<.., SP>, k_sp_*, k_contract_sp, k_contract_spg).

CPU: the identity itself on the oracle's matrices, and the Hilbert-curve station order (mik_station_order needs no GPU).
GPU: the sparse path against the oracle, against the dense path of the same library, and against the reference's stored answers
(every spherical fixture; the full-size config-5 slab), at BASELINE's tolerances |dz| <= 1e-8, |dsigma^2| <= 1e-6."""
import os

import numpy as np
import pytest

from oracle import kriging_oracle as ko
from tests import _fixtures as fx

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

Z_TOL, SS_TOL = 1e-8, 1e-6


# ---------------------------------------------------------------------------------------------- CPU
def test_identity_on_the_oracle_matrices():
    """delta = b + s u  =>  sigma^2 = 2 s - delta^T A^-1 delta and z = c . delta, with and without drift rows (uk.py:915-918: the last
    column of the matrix is [1_N; 0] there too)."""
    rng = np.random.default_rng(7)
    n = 400
    x, y = rng.random(n), rng.random(n)
    v = np.sin(6 * x) * np.cos(4 * y) + 0.1 * rng.standard_normal(n)
    par = [1.0, 0.25, 0.02]
    for rl in (False, True):
        st = ko.KrigingState(ndim=2, coords_orig=np.stack([x, y], 1), values=v, model="spherical",
                             params=ko.internal_parameters("spherical", par), regional_linear=rl)
        a = ko.kriging_matrix(st)
        ainv = np.linalg.inv(a)
        px, py = rng.random(50), rng.random(50)
        px[:3], py[:3] = x[:3], y[:3]  # exact hits: b_k = 0, delta_k = s
        zr, sr = ko.execute(st, "points", px, py)
        s = par[0]  # user parameters are [full sill, range, nugget]: s = psill + nugget
        m = a.shape[0]
        for t in range(50):
            d = np.hypot(x - px[t], y - py[t])
            gam = ko.variogram(st.model, st.params, d)
            b = np.zeros(m)
            b[:n] = -gam
            b[:n][d <= 1e-10] = 0.0
            if rl:
                b[n], b[n + 1] = px[t], py[t]
            b[m - 1] = 1.0
            delta = b.copy()
            delta[:n] += s
            assert np.all(delta[:n][d > par[1]] == 0.0)  # exactly zero beyond the range
            c = ainv[:, :n] @ v
            assert abs(c @ delta - zr[t]) < 1e-9
            assert abs(2 * s - delta @ ainv @ delta - sr[t]) < 1e-9


def test_station_order_is_a_hilbert_curve():
    """On a 2^k lattice consecutive stations of the order are lattice neighbours (2-D and 3-D); on random stations 16 consecutive
    ones are spatially compact; the order is a permutation and deterministic."""
    from pykrige_amd import _lib

    k = 16
    gx, gy = np.meshgrid(np.arange(k, dtype=float), np.arange(k, dtype=float))
    o = _lib.station_order(gx.ravel(), gy.ravel())
    assert sorted(o.tolist()) == list(range(k * k))
    step = np.abs(np.diff(gx.ravel()[o])) + np.abs(np.diff(gy.ravel()[o]))
    assert np.all(step == 1.0)
    k = 8
    g3 = np.stack(np.meshgrid(*[np.arange(k, dtype=float)] * 3, indexing="ij"), -1).reshape(-1, 3)
    o = _lib.station_order(g3[:, 0], g3[:, 1], g3[:, 2])
    assert sorted(o.tolist()) == list(range(k ** 3))
    assert np.all(np.abs(np.diff(g3[o], axis=0)).sum(1) == 1.0)
    rng = np.random.default_rng(5)
    x, y = rng.random(8000), rng.random(8000)
    o = _lib.station_order(x, y)
    assert np.array_equal(o, _lib.station_order(x, y))
    xs, ys = x[o].reshape(-1, 16), y[o].reshape(-1, 16)
    diam = np.hypot(xs.max(1) - xs.min(1), ys.max(1) - ys.min(1))
    assert np.median(diam) < 4.0 * np.sqrt(16 / 8000.0)  # a random 16-subset would span the unit square


def test_station_order_of_geographic_problems_is_compact_on_the_sphere():
    """coordinates_type = 'geographic' (round 5): the order is the plane curve through (lon, lat) -- lon / lat -> sphere is continuous, so
    16 consecutive stations are neighbours ON THE SPHERE too, also at the date line and the poles (where only the BOXES must not be
    lon / lat boxes: the library builds them from the unit vectors)."""
    from pykrige_amd import _lib

    rng = np.random.default_rng(12)
    n = 4096
    lon = rng.uniform(-180.0, 180.0, n)
    lat = np.degrees(np.arcsin(rng.uniform(-1.0, 1.0, n)))  # uniform on the sphere
    o = _lib.station_order(lon, lat, geographic=True)
    assert sorted(o.tolist()) == list(range(n))
    assert np.array_equal(o, _lib.station_order(lon, lat))
    lo, la = np.radians(lon[o]), np.radians(lat[o])
    u = np.stack([np.cos(la) * np.cos(lo), np.cos(la) * np.sin(lo), np.sin(la)], 1)
    ext = np.array([np.ptp(u[i:i + 16], axis=0).max() for i in range(0, n, 16)])  # largest box edge of a tile of 16 stations (chord units)
    # 4096 points on a sphere of area 4 pi: a compact patch of 16 has a diameter of ~ sqrt(16 * 4 pi / 4096) = 0.22
    assert np.median(ext) < 0.3 and np.quantile(ext, 0.9) < 0.4 and ext.max() < 0.7, (np.median(ext), np.quantile(ext, 0.9), ext.max())


def test_gathered_row_groups_reproduce_the_quadratic_form():
    """The tiling of k_contract_spg (option "sparse_rows" 16), restated in NumPy: the ascending list of a point block's active K tiles
    (16 stations each) is also the list of its active 16-row groups; tile r takes list entries [8r, 8r + 8) as its rows, walks the
    entries beyond them downwards as full K tiles, then its own groups as a triangle -- the group in list position j meets the K tiles
    of positions >= j, its accumulator doubled when the walk reaches its own 16 x 16 square.  The sum over the tiles is
    delta^T A delta over the active groups, i.e. over everything (the inactive groups hold exact zeros).  Also the order in which
    k_sp_tiles_g lays the tile records of a group of four point blocks out (tile position ascending, point block fast)."""
    rng = np.random.default_rng(3)
    ng_all, npts = 45, 24
    m = 16 * ng_all
    a = rng.standard_normal((m, m))
    a = a + a.T
    for nk in (1, 5, 8, 9, 16, 23, 45):
        active = np.sort(rng.choice(ng_all, nk, replace=False))
        delta = np.zeros((m, npts))
        for g in active:
            delta[16 * g:16 * g + 16] = rng.standard_normal((16, npts))
        want = np.einsum("it,ij,jt->t", delta, a, delta)
        got = np.zeros(npts)
        for r in range((nk + 7) // 8):
            g0, n = 8 * r, nk - 8 * r
            ngr = min(n, 8)
            acc = np.zeros((8, 16, npts))
            for w in range(n - 1, -1, -1):  # list positions relative to the tile's first group
                kt = active[g0 + w]
                dk = delta[16 * kt:16 * kt + 16]
                for gi in range(ngr):
                    rows = slice(16 * active[g0 + gi], 16 * active[g0 + gi] + 16)
                    if w >= 8:
                        acc[gi] += a[rows, 16 * kt:16 * kt + 16] @ dk
                    elif w >= gi:
                        if w == gi:
                            acc[gi] *= 2.0
                        acc[gi] += a[rows, 16 * kt:16 * kt + 16] @ dk
            for gi in range(ngr):
                got += np.einsum("it,it->t", delta[16 * active[g0 + gi]:16 * active[g0 + gi] + 16], acc[gi])
        assert np.allclose(got, want, rtol=1e-12, atol=1e-9), nk
    # record placement inside one group of MIK_ST = 4 point blocks: every thread computes its own tiles' slots
    for nr in ([3, 3, 3, 3], [5, 1, 0, 2], [1, 7, 7, 2], [0, 0, 4, 0]):
        order = [(r, q) for r in range(max(nr)) for q in range(4) if r < nr[q]]  # what one serial writer would produce
        slots = {}
        for q in range(4):
            w = 0
            for r in range(nr[q]):
                on = [1 if nr[qq] > r else 0 for qq in range(4)]
                slots[w + sum(on[:q])] = (r, q)
                w += sum(on)
        assert [slots[i] for i in range(len(order))] == order, nr


def test_point_sort_restated_in_numpy():
    """The device sort of the points (k_ps_hist / k_ps_scan / k_ps_scatter), restated: per launch segment, two passes of a stable counting
    sort over 10-bit digits -- digit counts per block of 4096 keys, exclusive scan digit-major / block-minor, every (wavefront = 1024
    consecutive keys, digit) gets a base, keys take ranks in index order.  Result = a stable argsort of the 20-bit keys inside every
    segment, whatever the segment length (no multiple of the block size, shorter than a block, a single key).  The keys themselves:
    the stations' Hilbert key on a 10-bit lattice -- a walk whose consecutive lattice points are neighbours."""
    rng = np.random.default_rng(8)
    DB, TILE = 10, 4096

    def radix_pass(key, idx, chunk, shift):
        npt = key.size
        kout, iout = np.empty_like(key), np.empty_like(idx)
        for seg in range((npt + chunk - 1) // chunk):
            lo, hi = seg * chunk, min(npt, (seg + 1) * chunk)
            bps = (chunk + TILE - 1) // TILE
            dig = (key[lo:hi] >> shift) & ((1 << DB) - 1)
            table = np.zeros((1 << DB, bps), dtype=np.int64)
            for b in range(bps):
                blk = dig[b * TILE:(b + 1) * TILE]
                table[:, b] = np.bincount(blk, minlength=1 << DB)
            base = np.concatenate([[0], np.cumsum(table.ravel())[:-1]]).reshape(table.shape)  # digit major, block minor
            for b in range(bps):
                run = base[:, b].copy()
                for w0 in range(b * TILE, min((b + 1) * TILE, hi - lo), 1024):  # a wavefront's 1024 keys, 64 at a time = index order
                    for t in range(w0, min(w0 + 1024, (b + 1) * TILE, hi - lo)):
                        d = dig[t]
                        kout[lo + run[d]], iout[lo + run[d]] = key[lo + t], idx[lo + t]
                        run[d] += 1
        return kout, iout

    for npt, chunk in ((10000, 4096 + 128), (3000, 131072), (1, 128), (9000, 2048)):
        key = rng.integers(0, 1 << 20, npt).astype(np.int64)
        key[::7] = key[0]  # ties: stability matters
        k1, i1 = radix_pass(key, np.arange(npt), chunk, 0)
        k2, i2 = radix_pass(k1, i1, chunk, DB)
        for seg in range((npt + chunk - 1) // chunk):
            lo, hi = seg * chunk, min(npt, (seg + 1) * chunk)
            want = lo + np.argsort(key[lo:hi], kind="stable")
            assert np.array_equal(i2[lo:hi], want), (npt, chunk, seg)
    from pykrige_amd import _lib

    k = 32  # a 2^5 lattice inside the 10-bit one: consecutive points of the order are lattice neighbours
    gx, gy = np.meshgrid(np.arange(k, dtype=float), np.arange(k, dtype=float))
    o = _lib.station_order(gx.ravel(), gy.ravel())
    assert np.all(np.abs(np.diff(gx.ravel()[o])) + np.abs(np.diff(gy.ravel()[o])) == 1.0)


# ---------------------------------------------------------------------------------------------- GPU
def _run(m, style, args, sparse, chunk=None, rows=None, lanes=2, **kw):
    h = m._get_handle()
    h.set_option("sparse", sparse)
    h.set_option("sparse_rows", -1 if rows is None else rows)  # 16 (default): gathered row groups, lists per 8 stations; 128: aligned blocks, lists per 16
    h.set_option("sparse_lanes", lanes)
    if chunk:
        h.set_option("chunk", chunk)
    z, ss = m.execute(style, *args, **kw)
    return np.ma.getdata(z), np.ma.getdata(ss), dict(m.last_timing)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["ok2d", "ok2d_short_range", "uk2d", "ok3d", "uk3d_aniso", "ok2d_noexact", "ok2d_aniso"])
def test_sparse_contraction_against_oracle_and_dense(case):
    import pykrige_amd as pa

    rng = np.random.default_rng(11)
    n = 1500
    if case in ("ok3d", "uk3d_aniso"):
        (x, y, zc), v = fx.synth(3, n, 3)
        axes = [np.linspace(0, 1, 23), np.linspace(0, 1, 19), np.linspace(0, 1, 11)]
        for k in range(6):
            x[k], y[k], zc[k] = axes[0][3 * k], axes[1][2 * k], axes[2][k]
        par = [1.0, 0.35, 0.02]
        if case == "ok3d":
            m = pa.OrdinaryKriging3D(x, y, zc, v, variogram_model="spherical", variogram_parameters=par)
            st = ko.KrigingState(ndim=3, coords_orig=np.stack([x, y, zc], 1), values=v, model="spherical",
                                 params=ko.internal_parameters("spherical", par), scaling=[1.0, 1.0], angle=[0.0, 0.0, 0.0])
        else:
            m = pa.UniversalKriging3D(x, y, zc, v, variogram_model="spherical", variogram_parameters=par, drift_terms=["regional_linear"],
                                      anisotropy_scaling_y=1.4, anisotropy_scaling_z=0.8, anisotropy_angle_x=10.0, anisotropy_angle_y=20.0,
                                      anisotropy_angle_z=30.0)
            st = ko.KrigingState(ndim=3, coords_orig=np.stack([x, y, zc], 1), values=v, model="spherical",
                                 params=ko.internal_parameters("spherical", par), scaling=[1.4, 0.8], angle=[10.0, 20.0, 30.0],
                                 regional_linear=True)
    else:
        (x, y), v = fx.synth(2, n, 2)
        axes = [np.linspace(0, 1, 161), np.linspace(0, 1, 53)]
        for k in range(8):
            x[k], y[k] = axes[0][15 * k + 1], axes[1][3 * k + 2]
        par = [1.0, 0.08, 0.0] if case == "ok2d_short_range" else [1.0, 0.2, 0.01]
        kw = dict(exact_values=False) if case == "ok2d_noexact" else {}
        okw = dict(exact_values=False) if case == "ok2d_noexact" else {}
        if case == "ok2d_aniso":
            kw.update(anisotropy_scaling=2.5, anisotropy_angle=35.0)
            okw.update(scaling=[2.5], angle=[35.0])
        if case == "uk2d":
            wells = [[0.3137, 0.7219, 1.0], [0.6621, 0.2483, -0.5]]
            m = pa.UniversalKriging(x, y, v, variogram_model="spherical", variogram_parameters=par,
                                    drift_terms=["regional_linear", "point_log"], point_drift=wells)
            st = ko.KrigingState(ndim=2, coords_orig=np.stack([x, y], 1), values=v, model="spherical",
                                 params=ko.internal_parameters("spherical", par), regional_linear=True, point_log=np.array(wells))
        else:
            m = pa.OrdinaryKriging(x, y, v, variogram_model="spherical", variogram_parameters=par, **kw)
            st = ko.KrigingState(ndim=2, coords_orig=np.stack([x, y], 1), values=v, model="spherical",
                                 params=ko.internal_parameters("spherical", par), **okw)
    zr, sr = ko.execute(st, "grid", *axes)
    zd, sd, td = _run(m, "grid", axes, 0)
    assert td["sparse"] == 0 and td["stations_sorted"] == 0
    # (lanes: 2 = the default since round 5 -- the launches alternate between two streams and two sets of work buffers --, 1 = one stream)
    for chunk, rows, lanes in ((131072, None, 2), (2048, None, 2), (131072, 128, 1), (2048, 16, 1), (4096, None, 2), (1024, None, 1), (1024, 128, 2)):
        zs, ss, ts = _run(m, "grid", axes, 1, chunk=chunk, rows=rows, lanes=lanes)
        assert ts["sparse"] == 1 and ts["stations_sorted"] == 1
        assert ts["sparse_rows"] == (rows or 16)  # gathered 16-row groups are the default
        assert ts["sparse_ktile"] == (16 if rows == 128 else 8)  # (aligned row blocks keep their 16-station lists)
        assert 0 < ts["sparse_tiles"] <= ts["sparse_tiles_dense"]
        assert np.abs(zs - zr).max() <= Z_TOL and np.abs(ss - sr).max() <= SS_TOL, (np.abs(zs - zr).max(), np.abs(ss - sr).max())
        assert np.abs(zs - zd).max() <= Z_TOL and np.abs(ss - sd).max() <= SS_TOL
        if rows == 128 and chunk == 131072:
            t128 = ts
        elif chunk == 131072:
            t8 = ts
    # gathered groups with pairs of 8-station tiles never execute more than aligned blocks with 16-station tiles do: fewer or equal
    # off-diagonal K steps and triangle products
    assert t8["sparse_ktiles"] <= t128["sparse_ktiles"] and t8["sparse_diag_products"] <= t128["sparse_diag_products"], (t8, t128)
    # two launch lanes (two streams, two sets of work buffers; the second lane waits for the sort of the points)
    m._get_handle().set_option("sort_points", 1)
    zs, ss, ts = _run(m, "grid", axes, 1, chunk=1024, lanes=2)
    m._get_handle().set_option("sort_points", -1)
    assert ts["sparse"] == 1 and ts["points_sorted"] == 1
    assert np.abs(zs - zr).max() <= Z_TOL and np.abs(ss - sr).max() <= SS_TOL, (np.abs(zs - zr).max(), np.abs(ss - sr).max())
    if case == "ok2d_short_range":
        assert ts["sparse_tiles"] < 0.6 * ts["sparse_tiles_dense"], ts  # range 0.08 of the unit square: most tiles are skipped
    # Hilbert-ordered stations with the dense contraction (option 2): the order alone changes nothing beyond rounding
    zo, so, to = _run(m, "grid", axes, 2)
    assert to["sparse"] == 0 and to["stations_sorted"] == 1
    assert np.abs(zo - zr).max() <= Z_TOL and np.abs(so - sr).max() <= SS_TOL
    # points and masked styles through the sparse path (points in random order: every block is active; still exact)
    pts = [rng.random(700) for _ in axes]
    zp, sp = ko.execute(st, "points", *pts)
    zs, ss, ts = _run(m, "points", pts, 1)
    assert ts["sparse"] == 1
    assert np.abs(zs - zp).max() <= Z_TOL and np.abs(ss - sp).max() <= SS_TOL
    if len(axes) == 2:
        mask = rng.random((axes[1].size, axes[0].size)) < 0.4
        zs, ss, ts = _run(m, "masked", axes, 1, mask=mask)
        assert ts["sparse"] == 1
        assert np.abs(zs - zr)[~mask].max() <= Z_TOL and np.abs(ss - sr)[~mask].max() <= SS_TOL


@pytest.mark.gpu
@pytest.mark.parametrize("ndim", [2, 3])
def test_sorted_points_make_any_point_order_compact(ndim):
    """Option "sort_points" (k_ps_*): the points of every launch are kriged in Hilbert-curve order among themselves, results come back
    in the caller's order.  A shuffled point list (every block of 128 consecutive points sees the whole domain) then costs what a
    sorted one costs; a 3-D grid's row segments become compact patches.  Same answers as the oracle and as the unsorted run."""
    import pykrige_amd as pa

    rng = np.random.default_rng(23)
    n, npts = 1500, 30000
    coords, v = fx.synth(6, n, ndim)
    par = [1.0, 0.1, 0.01] if ndim == 2 else [1.0, 0.25, 0.01]
    if ndim == 2:
        m = pa.OrdinaryKriging(coords[0], coords[1], v, variogram_model="spherical", variogram_parameters=par)
        st = ko.KrigingState(ndim=2, coords_orig=np.stack(coords, 1), values=v, model="spherical", params=ko.internal_parameters("spherical", par))
    else:
        m = pa.UniversalKriging3D(coords[0], coords[1], coords[2], v, variogram_model="spherical", variogram_parameters=par,
                                  drift_terms=["regional_linear"])
        st = ko.KrigingState(ndim=3, coords_orig=np.stack(coords, 1), values=v, model="spherical", params=ko.internal_parameters("spherical", par),
                             scaling=[1.0, 1.0], angle=[0.0, 0.0, 0.0], regional_linear=True)
    pts = [rng.random(npts) for _ in range(ndim)]
    for k in range(5):  # exact hits
        for d in range(ndim):
            pts[d][7 * k] = coords[d][k]
    zr, sr = ko.execute(st, "points", *pts)
    out = {}
    for sort, chunk in ((0, None), (1, None), (1, 4096 + 128)):  # (a chunk that is no multiple of the sort's 4096-key blocks)
        h = m._get_handle()
        h.set_option("sort_points", sort)
        z, ss, t = _run(m, "points", pts, 1, chunk=chunk or 131072)
        assert t["sparse"] == 1 and t["points_sorted"] == sort
        assert np.abs(z - zr).max() <= Z_TOL and np.abs(ss - sr).max() <= SS_TOL, (sort, chunk, np.abs(z - zr).max(), np.abs(ss - sr).max())
        out[(sort, chunk)] = t
    assert out[(1, None)]["sparse_ktiles"] < 0.5 * out[(0, None)]["sparse_ktiles"], out  # compact blocks: a fraction of the K tiles
    # a grid through the sorted path, every style's bookkeeping (masked: compacted points)
    axes = [np.linspace(0, 1, 57), np.linspace(0, 1, 41)] + ([np.linspace(0, 1, 13)] if ndim == 3 else [])
    zg, sg = ko.execute(st, "grid", *axes)
    m._get_handle().set_option("sort_points", 1)
    z, ss, t = _run(m, "grid", axes, 1, chunk=2048)
    assert t["points_sorted"] == 1
    assert np.abs(z - zg).max() <= Z_TOL and np.abs(ss - sg).max() <= SS_TOL
    if ndim == 2:
        mask = rng.random((axes[1].size, axes[0].size)) < 0.3
        z, ss, t = _run(m, "masked", axes, 1, mask=mask)
        assert np.abs(z - zg)[~mask].max() <= Z_TOL and np.abs(ss - sg)[~mask].max() <= SS_TOL


@pytest.mark.gpu
def test_sorted_points_carry_host_evaluated_drift_values():
    """Specified and functional drift terms are evaluated on the host, one value per point in the caller's order (uk.py:949-979); the
    sorted contraction reaches them through the permutation like the coordinates.  Sorted = unsorted = dense, 2-D and 3-D."""
    import pykrige_amd as pa

    rng = np.random.default_rng(57)
    (x, y), v = fx.synth(21, 900, 2)
    px, py = rng.random(6000), rng.random(6000)
    m2 = pa.UniversalKriging(x, y, v, variogram_model="spherical", variogram_parameters=[1.0, 0.25, 0.02],
                             drift_terms=["regional_linear", "specified", "functional"], specified_drift=[np.sin(3 * x) * y],
                             functional_drift=[lambda a, b: a * b])
    kw2 = dict(specified_drift_arrays=[np.sin(3 * px) * py])
    (x3, y3, z3), v3 = fx.synth(22, 900, 3)
    qx, qy, qz = rng.random(5000), rng.random(5000), rng.random(5000)
    m3 = pa.UniversalKriging3D(x3, y3, z3, v3, variogram_model="spherical", variogram_parameters=[1.0, 0.4, 0.02],
                               drift_terms=["specified"], specified_drift=[x3 * z3 + y3])
    kw3 = dict(specified_drift_arrays=[qx * qz + qy])
    for m, pts, kw in ((m2, [px, py], kw2), (m3, [qx, qy, qz], kw3)):
        zd, sd, td = _run(m, "points", pts, 0, **kw)
        out = {}
        for sort in (0, 1):
            m._get_handle().set_option("sort_points", sort)
            zs, ss, ts = _run(m, "points", pts, 1, chunk=2048, **kw)
            assert ts["sparse"] == 1 and ts["points_sorted"] == sort
            assert np.abs(zs - zd).max() <= Z_TOL and np.abs(ss - sd).max() <= SS_TOL, (sort, np.abs(zs - zd).max(), np.abs(ss - sd).max())
            out[sort] = (zs, ss)
        # z is a per-point sum over the stations; since round 5 k_rhs walks the point block's list of candidate tiles, so WHICH lane adds
        # which station follows the block the point falls into: equal to rounding across point orders, no longer bit for bit
        assert np.abs(out[0][0] - out[1][0]).max() <= 1e-13


@pytest.mark.gpu
def test_sorted_points_on_degenerate_point_lists():
    """The device sort of the points (k_ps_*) on lists it has little to work with: one point, 63 / 129 points (below and just above one
    block), all points identical (no extent: every key is 0, the stable sort is the identity), points on a line, a launch size of 128.
    Same answers as the dense contraction of the same library."""
    import pykrige_amd as pa

    rng = np.random.default_rng(41)
    (x, y), v = fx.synth(12, 700, 2)
    m = pa.OrdinaryKriging(x, y, v, variogram_model="spherical", variogram_parameters=[1.0, 0.2, 0.01])
    lists = {"one": (np.array([0.4]), np.array([0.6])), "63": (rng.random(63), rng.random(63)), "129": (rng.random(129), rng.random(129)),
             "identical": (np.full(300, 0.37), np.full(300, 0.52)), "line": (rng.random(500), np.full(500, 0.25)),
             "station": (np.full(200, x[3]), np.full(200, y[3]))}
    for name, pts in lists.items():
        zd, sd, td = _run(m, "points", list(pts), 0)
        for chunk in (131072, 128):
            m._get_handle().set_option("sort_points", 1)
            zs, ss, ts = _run(m, "points", list(pts), 1, chunk=chunk)
            assert ts["sparse"] == 1 and ts["points_sorted"] == 1, (name, ts)
            assert np.abs(zs - zd).max() <= Z_TOL and np.abs(ss - sd).max() <= SS_TOL, (name, chunk, np.abs(zs - zd).max(), np.abs(ss - sd).max())


@pytest.mark.gpu
def test_sparse_factor_is_handed_out_in_the_callers_order():
    """mik_get_matrix(1) un-permutes the Hilbert-ordered factor: equal to the dense path's inverse and to LAPACK's."""
    from tests.test_hip_parity import _handle_for

    (x, y), v = fx.synth(4, 500, 2)
    st = ko.KrigingState(ndim=2, coords_orig=np.stack([x, y], 1), values=v, model="spherical",
                         params=ko.internal_parameters("spherical", [1.0, 0.3, 0.02]), regional_linear=True)
    ref = np.linalg.inv(ko.kriging_matrix(st))
    for sparse in (0, 1):
        h = _handle_for(st, sparse=sparse)
        h.factor()
        got = h.get_matrix(1)
        assert np.abs(got - ref).max() <= 1e-9 * np.abs(ref).max()


@pytest.mark.gpu
@pytest.mark.parametrize("name", [n for n in fx.names() if "spherical" in n])
def test_sparse_path_on_the_reference_fixtures(name):
    """Every spherical execute fixture of the real reference through the sparse path (forced: they are small)."""
    g = fx.load(name)
    m = fx.amd_model_from(name, g)
    m._get_handle().set_option("sparse", 1)
    z, ss = m.execute("grid", *fx.grid_args(g))
    assert m.last_timing["sparse"] == 1
    scale = max(1.0, float(np.abs(g["z"]).max()))
    assert np.abs(np.ma.getdata(z) - g["z"]).max() <= Z_TOL * scale
    assert np.abs(np.ma.getdata(ss) - g["ss"]).max() <= SS_TOL * scale


@pytest.mark.gpu
def test_sparse_path_in_a_device_group():
    """A 3-member group (aliased onto the one GPU): the members sort their copies of the stations the same way, the broadcast
    factor is in that order; results equal one device's bit for bit while the points are kriged in the caller's order
    (sort_points = 0: the same blocks of 128 points whatever the slabs), and to rounding when every member puts the points of ITS
    launches in Hilbert order (sort_points = 1, the default: a block's partial sums follow the block's list of active stations)."""
    import pykrige_amd as pa

    (x, y), v = fx.synth(9, 1300, 2)
    axes = [np.linspace(0, 1, 140), np.linspace(0, 1, 37)]
    for sort in (0, 1):
        outs = []
        for ndev in (1, 3):
            m = pa.OrdinaryKriging(x, y, v, variogram_model="spherical", variogram_parameters=[1.0, 0.15, 0.01])
            h = m._get_handle()
            h.set_option("sparse", 1)
            h.set_option("sort_points", sort)
            if ndev > 1:
                h.set_devices(ndev, alias=True)
            z, ss = m.execute("grid", *axes)
            assert m.last_timing["sparse"] == 1 and m.last_timing["points_sorted"] == sort
            outs.append((np.ma.getdata(z).copy(), np.ma.getdata(ss).copy()))
        if sort == 0:
            assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
        else:
            assert np.abs(outs[0][0] - outs[1][0]).max() <= 1e-11 and np.abs(outs[0][1] - outs[1][1]).max() <= 1e-11


@pytest.mark.gpu
def test_range_aware_contraction_on_geographic_coordinates():
    """Round 5: spherical model + coordinates_type='geographic' (ok.py:634-640, 990-996) takes the range-aware contraction: candidates by
    boxes of the unit vectors against the chord of the range, delta from the great-circle distance exactly as the dense path computes it.
    2000 stations over a hemisphere, range 10 degrees, a 96 x 64 lon / lat grid + exact hits: the oracle's numbers, the dense path's
    numbers to rounding, and fewer than 30 % of the dense tiles."""
    import pykrige_amd as pa

    rng = np.random.default_rng(31)
    n = 2000
    lon = rng.uniform(-170.0, 10.0, n)
    lat = np.degrees(np.arcsin(rng.uniform(-0.3, 0.95, n)))
    v = np.sin(np.radians(lon) * 2) * np.cos(np.radians(lat) * 3) + 0.1 * rng.standard_normal(n)
    glon, glat = np.linspace(-170.0, 10.0, 96), np.linspace(-15.0, 70.0, 64)
    lon[:4], lat[:4] = glon[[3, 30, 60, 90]], glat[[2, 20, 40, 60]]
    par = [1.0, 10.0, 0.02]
    outs = {}
    for sparse in (1, 0):
        ok = pa.OrdinaryKriging(lon, lat, v, variogram_model="spherical", variogram_parameters=par, coordinates_type="geographic")
        ok._get_handle().set_option("sparse", sparse)
        z, ss = ok.execute("grid", glon, glat, backend="loop")
        tm = ok.last_timing
        assert tm["sparse"] == sparse
        if sparse:
            frac = tm["sparse_tiles"] / tm["sparse_tiles_dense"]
            assert frac < 0.3, frac
            assert tm["points_sorted"] == 1  # 6144 points: sorted along a curve through their unit vectors
        outs[sparse] = (np.ma.getdata(z).copy(), np.ma.getdata(ss).copy())
    st = ko.KrigingState(ndim=2, coords_orig=np.stack([lon, lat], 1), values=v, model="spherical", params=ko.internal_parameters("spherical", par),
                         geographic=True)
    zr, sr = ko.execute(st, "grid", glon, glat)
    for sparse in (1, 0):
        assert np.abs(outs[sparse][0] - zr).max() <= Z_TOL and np.abs(outs[sparse][1] - sr).max() <= SS_TOL
    assert np.abs(outs[1][0] - outs[0][0]).max() <= 1e-11 and np.abs(outs[1][1] - outs[0][1]).max() <= 1e-11
    zg = outs[1][0]
    assert abs(zg[2, 3] - v[0]) <= 1e-9 and abs(outs[1][1][2, 3]) <= 1e-9  # an exact hit: the value, sigma^2 = 0


@pytest.mark.gpu
def test_profiling_instantiation_of_the_range_aware_contraction_gives_the_same_answers():
    """MIK_SPG_PROF=1 (read once per process: a child process) runs k_contract_spg's profiling instantiation -- s_memtime sums per phase of the
    tile loop and per triangle step, printed per launch to stderr (profiles/r05_spg_tile_phases.txt) --: same z / sigma^2 as the oracle, and
    the report accounts for every tile of the launch."""
    import subprocess
    import sys

    code = r"""
import sys, numpy as np
sys.path.insert(0, %r)
import pykrige_amd as pa
from tests import _fixtures as fx
from oracle import kriging_oracle as ko
(x, y), v = fx.synth(2, 1500, 2)
par = [1.0, 0.2, 0.01]
m = pa.OrdinaryKriging(x, y, v, variogram_model="spherical", variogram_parameters=par)
st = ko.KrigingState(ndim=2, coords_orig=np.stack([x, y], 1), values=v, model="spherical", params=ko.internal_parameters("spherical", par))
axes = [np.linspace(0, 1, 161), np.linspace(0, 1, 53)]
z, ss = m.execute("grid", *axes)
zr, sr = ko.execute(st, "grid", *axes)
t = dict(m.last_timing)
print("RESULT", t["sparse"], t["sparse_ktile"], float(np.abs(np.ma.getdata(z) - zr).max()), float(np.abs(np.ma.getdata(ss) - sr).max()), int(t["sparse_tiles"]))
""" % ROOT
    env = dict(os.environ, MIK_SPG_PROF="1")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    res = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT")][0].split()
    assert res[1] == "1" and res[2] == "8", res
    assert float(res[3]) <= Z_TOL and float(res[4]) <= SS_TOL, res
    assert "spg triangle steps (cycles per visit)" in r.stderr and "off-diagonal K steps" in r.stderr, r.stderr[-2000:]
    import re
    tiles = [int(v) for v in re.findall(r"spg phases, wavefront 0: (\d+) tiles of the launch", r.stderr)]
    assert tiles and sum(tiles) == int(res[5]), (tiles, res)  # every tile of every launch went through the profiled loop
