"""mik_set_grid: the points of style='grid' / 'masked' are generated on the device from the axes (meshgrid order +
anisotropy adjustment of ok.py:863-885, ok3d.py:866-883, core.py:120-193).  They must be the coordinates the reference
builds on the host -- the |d| <= eps coincidence rule (ok.py:665) and everything downstream depend on them."""
import os

import numpy as np
import pytest

from tests import _fixtures as fx


def _host_points(axes, center, scaling, angle):
    from pykrige_amd import core

    if len(axes) == 2:
        gx, gy = np.meshgrid(axes[0], axes[1])
        p = np.stack((gx.ravel(), gy.ravel()), 1)
    else:
        gz, gy, gx = np.meshgrid(axes[2], axes[1], axes[0], indexing="ij")
        p = np.stack((gx.ravel(), gy.ravel(), gz.ravel()), 1)
    return core.adjust_for_anisotropy(p, center, scaling, angle)


def test_anisotropy_matrices_are_the_ones_adjust_for_anisotropy_multiplies_by():
    """Host arithmetic (no GPU): rot / stretch handed to the device reproduce core.adjust_for_anisotropy when applied with
    NumPy's own dot -- i.e. they are the reference's matrices (core.py:150-154, 166-187), entry for entry."""
    from pykrige_amd import core

    rng = np.random.default_rng(0)
    for nd, sc, an in ((2, [3.0], [45.0]), (2, [0.37], [-123.4]), (3, [1.5, 2.0], [10.0, 20.0, 30.0]), (3, [1.0, 1.0], [0.0, 0.0, 0.0])):
        X = rng.random((500, nd)) * 7 - 2
        c = rng.random(nd)
        rot, st = core.anisotropy_matrices(nd, sc, an)
        Y = np.dot(np.diag(st), np.dot(rot, (X - c[None, :]).T)).T + c[None, :]
        assert np.array_equal(Y, core.adjust_for_anisotropy(X, c, sc, an))


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["2d_iso", "2d_aniso", "2d_aniso_b", "3d_iso", "3d_aniso", "geographic"])
def test_device_grid_points_equal_the_host_meshgrid(case):
    from pykrige_amd import _lib, core

    rng = np.random.default_rng(7)
    nd = 3 if case.startswith("3d") else 2
    axes = [np.sort(rng.random(n)) * 10 - 3 for n in ((37, 29, 11) if nd == 3 else (213, 157))]
    center = list(rng.random(nd) * 4)
    sc, an = {"2d_iso": ([1.0], [0.0]), "2d_aniso": ([3.0], [45.0]), "2d_aniso_b": ([0.37], [-123.4]), "3d_iso": ([1.0, 1.0], [0.0] * 3),
              "3d_aniso": ([1.5, 2.0], [10.0, 20.0, 30.0]), "geographic": ([1.0], [0.0])}[case]
    c, v = fx.synth(1, 50, nd)
    h = _lib.Handle(0)
    h.set_problem(ndim=nd, xs=c[0], ys=c[1], zs=c[2] if nd == 3 else None, values=v, model_id=4, params=[1.0, 0.3, 0.0])
    if case == "geographic":
        h.set_grid(axes)
        want = _host_points(axes, [0.0] * nd, [1.0], [0.0])
        want = np.stack(np.meshgrid(axes[0], axes[1]), -1).reshape(-1, 2)
    else:
        rot, st = core.anisotropy_matrices(nd, sc, an)
        h.set_grid(axes, center, rot, st)
        want = _host_points(axes, center, sc, an)
    got = h.get_points(nd)
    assert got.shape == want.shape
    # differences in units of the last place of the coordinate SCALE (entries near zero come out of a cancellation: their own
    # ulp is meaningless).  The device accumulates each dot product k-ascending with FMAs; whether that reproduces np.dot bit
    # for bit depends on the host's BLAS kernel (it does in the build container, it does not on the GPU box's EPYC host: 14 % of
    # the rotated coordinates differ by one rounding) -- either way the difference is ~1e-15 of the extent, against the 1e-10
    # of the coincidence rule (ok.py:665).
    scale = np.spacing(np.abs(want).max(axis=0))[None, :]
    ulp = np.abs(got - want) / scale
    print("device grid vs host meshgrid (%s): %d of %d coordinates differ, worst %.2f ulp of the coordinate scale (%.1e absolute)"
          % (case, int((got != want).sum()), got.size, float(ulp.max()), float(np.abs(got - want).max())))
    assert ulp.max() <= 2.0
    if case in ("2d_iso", "3d_iso", "geographic"):
        assert np.array_equal(got, want)  # no rotation: (x - c) + c has one way to round
    # a mask compacts the sequence; a cell range takes a slab of it
    mask = rng.random(want.shape[0]) < 0.35
    if case != "geographic":
        h.set_grid(axes, center, rot, st, mask=mask)
        assert np.array_equal(h.get_points(nd), got[~mask])
        lo, cnt = 1234, 4321
        h.set_grid(axes, center, rot, st, mask=mask[lo:lo + cnt], cell_range=(lo, cnt))
        assert np.array_equal(h.get_points(nd), got[lo:lo + cnt][~mask[lo:lo + cnt]])
        h.set_devices(3, alias=True)  # device group: the slabs follow each other
        h.set_problem(ndim=nd, xs=c[0], ys=c[1], zs=c[2] if nd == 3 else None, values=v, model_id=4, params=[1.0, 0.3, 0.0])
        h.set_grid(axes, center, rot, st, mask=mask)
        assert np.array_equal(h.get_points(nd), got[~mask])
    h.close()


@pytest.mark.gpu
def test_the_list_of_unmasked_cells_is_built_on_the_device_in_meshgrid_order():
    """style='masked' through mik_set_grid: the byte mask is compacted by k_mask_count / k_mask_scan / k_mask_write (ascending
    cell order = np.nonzero(~mask), ok.py:700).  Sizes around the 4096-cell block of the kernels, runs of masked cells longer
    than a block, masks with nothing / everything masked, and the scatter of the results back through the list."""
    from pykrige_amd import _lib

    rng = np.random.default_rng(11)
    c, v = fx.synth(2, 40, 2)
    h = _lib.Handle(0)
    h.set_problem(ndim=2, xs=c[0], ys=c[1], zs=None, values=v, model_id=4, params=[1.0, 0.3, 0.0])
    for nx, ny in ((1, 1), (64, 64), (64, 65), (4095, 1), (1, 4097), (300, 211), (1000, 1037)):
        axes = [np.linspace(0.0, 1.0, nx), np.linspace(0.0, 1.0, ny)]
        full = np.stack(np.meshgrid(axes[0], axes[1]), -1).reshape(-1, 2)
        ncell = nx * ny
        masks = [rng.random(ncell) < 0.5, rng.random(ncell) < 0.999, rng.random(ncell) < 0.001, np.zeros(ncell, bool), np.ones(ncell, bool)]
        runs = np.zeros(ncell, bool)
        runs[ncell // 7: ncell // 7 + 9000] = True  # more than two whole blocks with no unmasked cell
        runs[-1:] = True
        masks.append(runs)
        for m in masks:
            h.set_grid(axes, mask=m)
            assert int(h._lib.mik_points_resident(h._h)) == int((~m).sum())
            if (~m).any():
                assert np.array_equal(h.get_points(2), full[~m])
            h.factor()
            h.predict()
            z, ss = h.get_results()
            assert z.shape == (ncell,) and np.all(z[m] == 0.0) and np.all(ss[m] == 0.0)
            if (~m).any():
                h.set_points(full[~m, 0].copy(), full[~m, 1].copy())
                h.predict()
                z2, ss2 = h.get_results()
                assert np.array_equal(z[~m], z2) and np.array_equal(ss[~m], ss2)
    h.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["ok2d_exponential_exact", "ok2d_spherical_noexact", "ok3d_gaussian_aniso", "uk2d_rl_pl_node",
                                  "uk2d_external_z", "ok2d_masked_points", "ok2d_n2000", "uk3d_rl_func"])
def test_execute_from_axes_equals_execute_from_host_points(name):
    """execute('grid' / 'masked') through mik_set_grid against the same call through the host meshgrid (MIK_DEVICE_GRID=0) and
    against the reference's stored answer."""
    g = fx.load(name)
    style = "masked" if "mask" in g else "grid"
    kw = dict(mask=g["mask"].astype(bool)) if "mask" in g else {}
    if "spec_grid" in g:
        kw["specified_drift_arrays"] = [g["spec_grid"]]
    outs = []
    for dev_grid in ("1", "0"):
        os.environ["MIK_DEVICE_GRID"] = dev_grid
        try:
            m = fx.amd_model_from(name, g)
            outs.append(m.execute(style, *fx.grid_args(g), backend="vectorized", **kw))
        finally:
            del os.environ["MIK_DEVICE_GRID"]
    (za, sa), (zb, sb) = outs
    za, sa, zb, sb = (np.ma.getdata(a) for a in (za, sa, zb, sb))
    print("%s: device grid vs host points max|dz| %.2e max|dss| %.2e" % (name, np.abs(za - zb).max(), np.abs(sa - sb).max()))
    assert np.abs(za - zb).max() <= 1e-12 and np.abs(sa - sb).max() <= 1e-12
    if "z" in g and style == "grid":
        assert np.abs(za - g["z"]).max() <= 1e-8 and np.abs(sa - g["ss"]).max() <= 1e-6


@pytest.mark.gpu
def test_grid_cell_ranges_tile_the_whole_grid():
    """One rank's slab of a sharded grid (pykrige_amd.dist): cell ranges of mik_set_grid put together = the whole grid."""
    from pykrige_amd import _lib, core

    g = fx.load("ok2d_n2000")
    import pykrige_amd as pa

    ok = pa.OrdinaryKriging(g["x"], g["y"], g["v"], variogram_model="exponential", variogram_parameters=[1.0, 0.3, 0.0],
                            anisotropy_scaling=2.0, anisotropy_angle=30.0)
    z, ss = ok.execute("grid", g["gridx"], g["gridy"], backend="loop")
    h = ok._get_handle()
    P = ok._prepare("grid", (g["gridx"], g["gridy"]), None)
    parts = []
    n = P.npt
    for lo, hi in ((0, n // 3), (n // 3, n // 3 + 1), (n // 3 + 1, n)):
        P.load(h, 2, cell_range=(lo, hi - lo))
        h.predict()
        parts.append(h.get_results())
    assert np.array_equal(np.concatenate([p[0] for p in parts]), z.ravel())
    assert np.array_equal(np.concatenate([p[1] for p in parts]), ss.ravel())


@pytest.mark.gpu
def test_results_without_the_last_copy_are_the_same_results():
    """mik_take_results: the arrays execute() returns ARE the page-locked landing zone (no copy).  Same numbers as the copying
    path; they stay valid and unchanged through later calls on the same object and after the object is gone; the buffer of a
    dropped result is recycled."""
    import gc

    import pykrige_amd as pa

    g = fx.load("ok2d_n2000")
    ok = pa.OrdinaryKriging(g["x"], g["y"], g["v"], variogram_model="exponential", variogram_parameters=[1.0, 0.3, 0.0])
    z1, s1 = ok.execute("grid", g["gridx"], g["gridy"], backend="loop")
    assert not z1.flags.owndata  # a view of the library's buffer
    os.environ["MIK_ZERO_COPY"] = "0"
    try:
        zc, sc = ok.execute("grid", g["gridx"], g["gridy"], backend="loop")
    finally:
        del os.environ["MIK_ZERO_COPY"]
    assert zc.base is None or zc.base.flags.owndata
    assert np.array_equal(z1, zc) and np.array_equal(s1, sc)
    keep_z, keep_s = z1.copy(), s1.copy()
    addr1 = z1.__array_interface__["data"][0]
    z2, s2 = ok.execute("grid", g["gridx"][::-1].copy(), g["gridy"], backend="loop")  # another call: must not touch z1 / s1
    assert z2.__array_interface__["data"][0] != addr1
    assert np.array_equal(z1, keep_z) and np.array_equal(s1, keep_s)
    assert np.array_equal(z2[:, ::-1], keep_z)
    del ok
    gc.collect()
    assert np.array_equal(z1, keep_z) and np.array_equal(s1, keep_s)  # the handle is gone, the arrays are not
    # dropped results' buffers come back: a loop that drops what it gets cycles through a bounded set of buffers
    del z2, s2
    gc.collect()
    ok2 = pa.OrdinaryKriging(g["x"], g["y"], g["v"], variogram_model="exponential", variogram_parameters=[1.0, 0.3, 0.0])
    seen = set()
    for _ in range(8):
        z3, s3 = ok2.execute("grid", g["gridx"], g["gridy"], backend="loop")
        seen.add(z3.__array_interface__["data"][0])
        assert np.array_equal(z3, keep_z)
    assert len(seen) <= 3, seen  # (the one in flight, the one held by the caller, the one being readied)
    h = ok2._get_handle()
    a, b = h.get_results()  # asked twice before the next predict: the same arrays again
    assert a.__array_interface__["data"][0] == z3.__array_interface__["data"][0]
    # masked style and device groups keep the copying path (scatter through the mask / one slab per device)
    m = np.zeros((g["gridy"].size, g["gridx"].size), dtype=bool)
    m[::3, ::2] = True
    zm, sm = ok2.execute("masked", g["gridx"], g["gridy"], mask=m, backend="loop")
    assert np.array_equal(np.ma.getdata(zm)[~m], keep_z[~m]) and np.all(np.ma.getdata(zm)[m] == 0.0)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["2d_iso", "2d_aniso", "3d_aniso"])
def test_device_adjusted_points_equal_the_host_adjustment(case):
    """mik_adjust_points (round 3): coordinates uploaded raw by mik_set_points and adjusted in place on the device are the ones
    core.adjust_for_anisotropy (core.py:120-193) computes on the host -- to the last place of the coordinate scale, like the
    device-generated grids; with a mask (compaction before the adjustment) and through a 3-member group as well."""
    from pykrige_amd import _lib, core

    rng = np.random.default_rng(11)
    nd = 3 if case.startswith("3d") else 2
    n = 5000
    P = rng.random((n, nd)) * 9 - 2
    center = list(rng.random(nd) * 4)
    sc, an = {"2d_iso": ([1.0], [0.0]), "2d_aniso": ([0.37], [-123.4]), "3d_aniso": ([1.5, 2.0], [10.0, 20.0, 30.0])}[case]
    want = core.adjust_for_anisotropy(P.copy(), center, sc, an)
    rot, st = core.anisotropy_matrices(nd, sc, an)
    c, v = fx.synth(1, 50, nd)
    for members, mask in ((1, None), (1, rng.random(n) < 0.3), (3, None)):
        h = _lib.Handle(0)
        if members > 1:
            h.set_devices(members, alias=True)
        h.set_problem(ndim=nd, xs=c[0], ys=c[1], zs=c[2] if nd == 3 else None, values=v, model_id=4, params=[1.0, 0.3, 0.0])
        h.set_points(P[:, 0], P[:, 1], P[:, 2] if nd == 3 else None, mask=mask)
        h.adjust_points(center, rot, st)
        got = h.get_points(nd)
        ref = want if mask is None else want[~mask]
        scale = np.spacing(np.abs(ref).max(axis=0))[None, :]
        assert got.shape == ref.shape
        assert (np.abs(got - ref) / scale).max() <= 2.0
        if case == "2d_iso":
            assert np.array_equal(got, ref)
        h.close()
    # a grid's points are generated adjusted: the call refuses them
    h = _lib.Handle(0)
    h.set_problem(ndim=nd, xs=c[0], ys=c[1], zs=c[2] if nd == 3 else None, values=v, model_id=4, params=[1.0, 0.3, 0.0])
    h.set_grid([np.linspace(0, 1, 5)] * nd, center, rot, st)
    with pytest.raises(Exception):
        h.adjust_points(center, rot, st)
    h.close()


@pytest.mark.gpu
def test_execute_points_is_the_same_with_device_and_host_adjustment(monkeypatch):
    """style='points' (and every host-built point list): execute() with the coordinates adjusted on the device equals execute()
    with MIK_DEVICE_POINTS=0 (host adjustment, as the reference does at ok.py:879-885) on anisotropic fixtures, 2-D and 3-D, UK
    with specified drift arrays included."""
    import pykrige_amd as pa

    rng = np.random.default_rng(3)
    n = 150
    x, y, zc = rng.random(n) * 10, rng.random(n) * 6, rng.random(n) * 3
    v = np.sin(x) + 0.3 * y + 0.1 * rng.standard_normal(n)
    px, py, pz = rng.random(400) * 10, rng.random(400) * 6, rng.random(400) * 3
    px[:5], py[:5], pz[:5] = x[:5], y[:5], zc[:5]  # exact hits must stay exact hits
    models = [
        (pa.OrdinaryKriging(x, y, v, variogram_model="exponential", variogram_parameters=[1.0, 4.0, 0.01], anisotropy_scaling=2.5,
                            anisotropy_angle=33.0), (px, py), {}),
        (pa.UniversalKriging(x, y, v, variogram_model="spherical", variogram_parameters=[1.0, 5.0, 0.0], anisotropy_scaling=0.6,
                             anisotropy_angle=-71.0, drift_terms=["regional_linear", "specified"], specified_drift=[x * y]),
         (px, py), {"specified_drift_arrays": [px * py]}),
        (pa.OrdinaryKriging3D(x, y, zc, v, variogram_model="gaussian", variogram_parameters=[1.0, 3.0, 0.05], anisotropy_scaling_y=1.5,
                              anisotropy_scaling_z=0.7, anisotropy_angle_x=10.0, anisotropy_angle_y=20.0, anisotropy_angle_z=30.0),
         (px, py, pz), {}),
    ]
    for m, pts, kw in models:
        monkeypatch.setenv("MIK_DEVICE_POINTS", "1")
        z1, s1 = m.execute("points", *pts, **kw)
        monkeypatch.setenv("MIK_DEVICE_POINTS", "0")
        z0, s0 = m.execute("points", *pts, **kw)
        np.testing.assert_allclose(np.ma.getdata(z1), np.ma.getdata(z0), rtol=0, atol=1e-11)
        np.testing.assert_allclose(np.ma.getdata(s1), np.ma.getdata(s0), rtol=0, atol=1e-11)
        assert np.abs(np.ma.getdata(s1)[:5]).max() <= 1e-9 or m.variogram_model_parameters[-1] > 0  # exact hits (zero nugget)


@pytest.mark.gpu
def test_empty_cell_range_is_empty_not_the_whole_grid():
    """mik_grid.cell_count = 0 is an EMPTY range (a rank of a sharded run with more ranks than cells), -1 the whole grid (round-3
    advisor finding: 0 used to mean the whole grid, and the caller's zero-length result arrays were overrun)."""
    from pykrige_amd import _lib

    rng = np.random.default_rng(3)
    x, y, v = rng.random(40), rng.random(40), rng.random(40)
    h = _lib.Handle(0)
    h.set_problem(ndim=2, xs=x, ys=y, zs=None, values=v, model_id=_lib.MODEL_IDS["exponential"], params=[0.9, 0.3, 0.1])
    h.factor()
    gx, gy = np.linspace(0, 1, 7), np.linspace(0, 1, 5)
    h.set_grid((gx, gy))
    h.predict()
    zall, sall = (a.copy() for a in h.get_results())
    assert zall.size == 35
    for mask in (None, np.zeros(0, dtype=bool)):
        h.set_grid((gx, gy), cell_range=(35, 0), mask=mask)
        assert h._lib.mik_points_resident(h._h) == 0
        h.predict()
        z, ss = h.get_results()
        assert z.size == 0 and ss.size == 0
    h.set_grid((gx, gy), cell_range=(10, 5))
    h.predict()
    z, ss = h.get_results()
    assert np.array_equal(z, zall[10:15]) and np.array_equal(ss, sall[10:15])
    h.close()


@pytest.mark.gpu
def test_adjusting_the_resident_points_twice_is_refused_and_stale_results_are_not_handed_out():
    """Round-3 advisor findings: a second mik_adjust_points on the same points would apply the anisotropy transform twice (now
    MIK_ESTATE); Handle.get_results after new points were set must not return the previous predict's arrays."""
    from pykrige_amd import _lib

    rng = np.random.default_rng(4)
    x, y, v = rng.random(30), rng.random(30), rng.random(30)
    h = _lib.Handle(0)
    h.set_problem(ndim=2, xs=x, ys=y, zs=None, values=v, model_id=_lib.MODEL_IDS["exponential"], params=[0.9, 0.3, 0.1])
    h.factor()
    px, py = rng.random(50), rng.random(50)
    h.set_points(px, py)
    rot = np.array([[np.cos(0.3), -np.sin(0.3)], [np.sin(0.3), np.cos(0.3)]])
    h.adjust_points([0.5, 0.5], rot, [1.0, 2.0])
    with pytest.raises(RuntimeError):
        h.adjust_points([0.5, 0.5], rot, [1.0, 2.0])
    h.predict()
    z1, _ = h.get_results()
    h.set_points(px[:20], py[:20])  # new points: the old results are gone
    with pytest.raises(RuntimeError):
        h.get_results()
    h.adjust_points([0.5, 0.5], rot, [1.0, 2.0])  # fresh points may be adjusted again
    h.predict()
    z2, _ = h.get_results()
    assert z2.size == 20 and np.array_equal(z2, z1[:20])
    h.close()


@pytest.mark.gpu
def test_page_locked_results_on_loan_are_capped():
    """mik_take_results lends the page-locked landing zone to the caller; beyond MIK_PIN_LENT_CAP bytes on loan it refuses and the
    copying mik_get_results serves the call (round-3 advisor finding: a caller keeping many results pinned without limit)."""
    import subprocess
    import sys

    code = '''
import os, sys
sys.path.insert(0, %r)
import numpy as np
from pykrige_amd import _lib
rng = np.random.default_rng(1)
x, y, v = rng.random(40), rng.random(40), rng.random(40)
h = _lib.Handle(0)
h.set_problem(ndim=2, xs=x, ys=y, zs=None, values=v, model_id=_lib.MODEL_IDS["exponential"], params=[0.9, 0.3, 0.1])
h.factor()
h.set_points(rng.random(4096), rng.random(4096))
kept, owned = [], 0
for i in range(6):
    h.predict()
    z, ss = h.get_results()
    kept.append((z, ss))
    owned += int(z.base is not None and not z.flags.owndata)   # a view of the lent buffer
assert all(np.array_equal(kept[0][0], k[0]) for k in kept)
print("LENT", owned)
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # each result = 2 x 4096 doubles = 64 KiB; a cap of 200 000 bytes allows three on loan
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=dict(os.environ, MIK_PIN_LENT_CAP="200000"))
    assert r.returncode == 0, r.stderr[-800:]
    lent = int([ln for ln in r.stdout.splitlines() if ln.startswith("LENT")][-1].split()[1])
    assert 1 <= lent <= 3, lent
