"""world_size-2 test of the multi-GPU host path on CPU (gloo): slab partition, unique-id exchange with the
RCCL-unavailable fallback, per-rank solve, gather.  The device handle is replaced by a stand-in that
computes with the CPU oracle (tests may do that; the product never does)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_slab_bounds_cover_everything():
    from pykrige_amd.dist import slab_bounds

    for n in (0, 1, 7, 8, 1000, 1001):
        for world in (1, 2, 3, 8):
            spans = [slab_bounds(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        slab_bounds(10, 2, 2)


class OracleHandle:
    """Stand-in for _lib.Handle with the same methods, backed by the CPU oracle."""

    def __init__(self):
        self.calls = []

    def set_problem(self, **kw):
        from oracle import kriging_oracle as ko

        self.kw = kw
        nd = kw["ndim"]
        coords = np.stack([kw["xs"], kw["ys"]] + ([kw["zs"]] if nd == 3 else []), 1)
        inv_model = {v: k for k, v in __import__("pykrige_amd")._lib.MODEL_IDS.items()}[kw["model_id"]]
        self.st = ko.KrigingState(ndim=nd, coords_orig=coords, values=kw["values"], model=inv_model,
                                  params=list(kw["params"]), scaling=[1.0] * (nd - 1), angle=[0.0] * (2 * nd - 3),
                                  exact_values=kw["exact_values"], regional_linear=kw["regional_linear"])
        self.calls.append("set_problem")

    def factor(self):
        self.calls.append("factor")

    def comm_init(self, world, rank, uid):
        raise RuntimeError("no RCCL on a CPU box")

    def bcast_factor(self, root):
        raise AssertionError("must not be reached after the RCCL fallback")

    def set_points(self, px, py, pz=None, mask=None, extra_rows=None):
        self.pts = np.stack([px, py] + ([pz] if pz is not None else []), 1)
        self.mask = mask

    def adjust_points(self, center, rot, stretch):
        """mik_adjust_points restated with NumPy: the anisotropy transform of coordinates set_points took raw."""
        c = np.asarray(center, float)[None, :]
        self.pts = (np.diag(stretch) @ (np.asarray(rot) @ (self.pts - c).T)).T + c
        self.calls.append("adjust_points")

    def set_grid(self, axes, center=None, rot=None, stretch=None, mask=None, extra_rows=None, cell_range=None):
        """mik_set_grid restated with NumPy: the reference's flattened meshgrid, the anisotropy transform, a cell range."""
        if len(axes) == 2:
            gx, gy = np.meshgrid(axes[0], axes[1])
            p = np.stack((gx.ravel(), gy.ravel()), 1)
        else:
            gz, gy, gx = np.meshgrid(axes[2], axes[1], axes[0], indexing="ij")
            p = np.stack((gx.ravel(), gy.ravel(), gz.ravel()), 1)
        if rot is not None:
            c = np.asarray(center, float)[None, :]
            p = (np.diag(stretch) @ (np.asarray(rot) @ (p - c).T)).T + c
        if cell_range is not None:
            p = p[cell_range[0]:cell_range[0] + cell_range[1]]
        self.pts, self.mask = p, mask
        self.calls.append("set_grid")

    def predict(self):
        self.calls.append("predict")
        self.window = None

    def predict_moving_window(self, k):
        self.calls.append("predict_moving_window")
        self.window = k

    def get_results(self):
        from oracle import kriging_oracle as ko

        z, ss = np.zeros(len(self.pts)), np.zeros(len(self.pts))
        keep = np.ones(len(self.pts), bool) if self.mask is None else ~np.asarray(self.mask, bool)
        if keep.any():
            if getattr(self, "window", None):
                z[keep], ss[keep] = ko.solve_points_moving_window(self.st, self.pts[keep], self.window)
            else:
                z[keep], ss[keep] = ko.solve_points(self.st, self.pts[keep])
        return z, ss


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    import pykrige_amd as pa
    from oracle import kriging_oracle as ko
    from pykrige_amd.dist import ShardedExecutor

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(5)
        x, y, v = rng.random(60), rng.random(60), rng.random(60)
        ok = pa.OrdinaryKriging(x, y, v, variogram_model="exponential", variogram_parameters=[1.0, 0.3, 0.01])
        hdl = OracleHandle()
        ex = ShardedExecutor(ok, group=dist.group.WORLD, handle_factory=lambda: hdl)  # a torch group is accepted when passed
        assert ex.exchange.startswith("redundant_factor"), ex.exchange  # RCCL init failed on every rank -> fallback
        gx, gy = np.linspace(0, 1, 13), np.linspace(0, 1, 9)
        z, ss = ex.execute("grid", gx, gy, backend="loop")
        mask = rng.random((9, 13)) < 0.3
        zm, ssm = ex.execute("masked", gx, gy, mask=mask, backend="loop")
        st = ko.KrigingState(ndim=2, coords_orig=np.stack([x, y], 1), values=v, model="exponential",
                             params=ko.internal_parameters("exponential", [1.0, 0.3, 0.01]))
        zr, sr = ko.execute(st, "grid", gx, gy)
        ok_grid = bool(np.allclose(z, zr, atol=1e-12) and np.allclose(ss, sr, atol=1e-12) and z.shape == (9, 13))
        ok_mask = bool(np.allclose(np.ma.getdata(zm)[~mask], zr[~mask], atol=1e-12) and np.all(np.ma.getdata(zm)[mask] == 0.0)
                       and isinstance(zm, np.ma.MaskedArray))
        # style='points' on an anisotropic model: every rank uploads its slab RAW and the handle adjusts it (mik_adjust_points)
        oka = pa.OrdinaryKriging(x, y, v, variogram_model="exponential", variogram_parameters=[1.0, 0.3, 0.01], anisotropy_scaling=2.0,
                                 anisotropy_angle=30.0)
        hda = OracleHandle()
        exa = ShardedExecutor(oka, group=dist.group.WORLD, handle_factory=lambda: hda)
        ppx, ppy = rng.random(37), rng.random(37)
        zp, ssp = exa.execute("points", ppx, ppy, backend="loop")
        sta = ko.KrigingState(ndim=2, coords_orig=np.stack([x, y], 1), values=v, model="exponential",
                              params=ko.internal_parameters("exponential", [1.0, 0.3, 0.01]), scaling=[2.0], angle=[30.0])
        zpr, spr = ko.execute(sta, "points", ppx, ppy)
        ok_pts = bool(np.allclose(zp, zpr, atol=1e-12) and np.allclose(ssp, spr, atol=1e-12) and hda.calls.count("adjust_points") == 1)
        q.put((rank, ok_grid, ok_mask and ok_pts, len(hdl.pts), hdl.calls.count("factor")))
    finally:
        dist.destroy_process_group()


def _sock_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import pykrige_amd as pa
    from oracle import kriging_oracle as ko
    from pykrige_amd.dist import ShardedExecutor, SocketGroup

    pg = SocketGroup(rank=rank, world=world, addr="127.0.0.1", port=port)
    try:
        assert pg.all_gather_object(rank * 10) == [0, 10, 20][:world]
        assert pg.broadcast_object(b"x" * 128 if rank == 0 else None) == b"x" * 128
        assert pg.all_reduce_max(1.0 + rank) == float(world)
        pg.barrier()
        rng = np.random.default_rng(6)
        x, y, zc, v = rng.random(50), rng.random(50), rng.random(50), rng.random(50)
        k3 = pa.OrdinaryKriging3D(x, y, zc, v, variogram_model="spherical", variogram_parameters=[1.0, 0.7, 0.05])
        hdl = OracleHandle()
        ex = ShardedExecutor(k3, group=pg, handle_factory=lambda: hdl)
        assert ex.exchange.startswith("redundant_factor")
        g = [np.linspace(0, 1, 5), np.linspace(0, 1, 4), np.linspace(0, 1, 3)]
        z, ss = ex.execute("grid", *g, backend="loop")
        st = ko.KrigingState(ndim=3, coords_orig=np.stack([x, y, zc], 1), values=v, model="spherical",
                             params=ko.internal_parameters("spherical", [1.0, 0.7, 0.05]), scaling=[1.0, 1.0], angle=[0.0] * 3)
        zr, sr = ko.execute(st, "grid", *g)
        zw, sw = ex.execute("grid", *g, backend="loop", n_closest_points=7)  # sharded moving window: no factor at all
        zwr, swr = ko.solve_points_moving_window(st, np.stack([a.ravel() for a in np.meshgrid(g[2], g[1], g[0], indexing="ij")][::-1], 1), 7)
        okw = np.allclose(zw.ravel(), zwr, atol=1e-12) and np.allclose(sw.ravel(), swr, atol=1e-12) and hdl.calls[-1] == "predict_moving_window" \
            and hdl.calls.count("factor") == 1
        npts3 = len(hdl.pts)
        # more ranks than cells (round-3 advisor finding): a 2-cell grid on 3 ranks leaves the last rank an EMPTY slab -- cell_range
        # (lo, 0) must mean "nothing", not "the whole grid"
        z2, ss2 = ex.execute("grid", np.array([0.25, 0.75]), np.array([0.5]), np.array([0.5]), backend="loop")
        z2r, s2r = ko.execute(st, "grid", np.array([0.25, 0.75]), np.array([0.5]), np.array([0.5]))
        ok2 = z2.shape == (1, 1, 2) and np.allclose(z2, z2r, atol=1e-12) and np.allclose(ss2, s2r, atol=1e-12) and len(hdl.pts) == (1 if rank < 2 else 0)
        q.put((rank, bool(ok2 and okw and np.allclose(z, zr, atol=1e-12) and np.allclose(ss, sr, atol=1e-12) and z.shape == (3, 4, 5)), npts3))
    finally:
        pg.close()


def test_socket_group_world3_sharded_execute():
    """The torch-free host process group (used by bench.py under torchrun): 3 ranks, collectives + a sharded 3-D execute."""
    import multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sock_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert [o[0] for o in out] == [0, 1, 2] and all(o[1] for o in out)
    assert sum(o[2] for o in out) == 5 * 4 * 3 and max(o[2] for o in out) - min(o[2] for o in out) <= 1


def test_sharded_execute_world2_gloo():
    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert [o[0] for o in out] == [0, 1]
    assert all(o[1] and o[2] for o in out), out
    assert sum(o[3] for o in out) == 13 * 9  # the last execute's slabs partition the grid
    assert all(o[4] == 2 for o in out)       # redundant factorisation: each rank factored in each execute


def _sock8_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    from pykrige_amd.dist import SocketGroup, slab_bounds

    pg = SocketGroup(rank=rank, world=world, addr="127.0.0.1", port=port)
    try:
        got = pg.all_gather_object((rank, slab_bounds(1000003, world, rank)))
        uid = pg.broadcast_object(bytes(range(128)) if rank == 0 else None)
        t = pg.all_reduce_max(0.25 * rank)
        for _ in range(3):
            pg.barrier()
        q.put((rank, got == [(r, slab_bounds(1000003, world, r)) for r in range(world)] and uid == bytes(range(128))
               and t == 0.25 * (world - 1)))
    finally:
        pg.close()


def test_socket_group_with_eight_ranks():
    """The rendezvous bench.py uses under `torch.distributed.run --nproc-per-node 8`: eight processes, the collectives
    the benchmark needs (gather, the 128-byte id broadcast, max-reduce, barriers)."""
    import multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sock8_worker, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert [o[0] for o in out] == list(range(8)) and all(o[1] for o in out)


class RcclStandInHandle(OracleHandle):
    """A stand-in whose communicator comes up: drives ShardedExecutor's rccl_bcast control flow on CPU.  The 'broadcast' is
    checked for its protocol: only rank 0 factors, every rank calls bcast_factor exactly once per execute, and nobody
    enters it when rank 0's factorisation failed."""

    def __init__(self, rank, fail_factor=False):
        super().__init__()
        self.rank, self.fail_factor = rank, fail_factor

    def comm_init(self, world, rank, uid):
        assert len(uid) == 128 and rank == self.rank
        self.calls.append("comm_init")

    def factor(self):
        if self.fail_factor:
            raise np.linalg.LinAlgError("singular matrix")
        super().factor()

    def bcast_factor(self, root):
        assert root == 0
        self.calls.append("bcast_factor")

    def factor_checksum(self):
        # what every rank's device would report for its copy of the inverse: equal everywhere unless `corrupt_rank` says
        # this rank's copy arrived damaged
        return (1, 2, 3, 4) if getattr(self, "corrupt_rank", None) != self.rank else (1, 2, 3, 5)


def _rccl_flow_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import pykrige_amd as pa
    from oracle import kriging_oracle as ko
    from pykrige_amd import _lib
    from pykrige_amd.dist import ShardedExecutor, SocketGroup

    _lib.Handle.comm_unique_id = staticmethod(lambda: bytes(range(128)))  # no RCCL on a CPU box: a fixed id
    pg = SocketGroup(rank=rank, world=world, addr="127.0.0.1", port=port, token="job-1")
    try:
        rng = np.random.default_rng(7)
        x, y, v = rng.random(40), rng.random(40), rng.random(40)
        ok = pa.OrdinaryKriging(x, y, v, variogram_model="exponential", variogram_parameters=[1.0, 0.3, 0.01])
        hdl = RcclStandInHandle(rank)
        ex = ShardedExecutor(ok, group=pg, handle_factory=lambda: hdl)
        assert ex.exchange == "rccl_bcast", ex.exchange
        gx, gy = np.linspace(0, 1, 7), np.linspace(0, 1, 6)
        z, ss = ex.execute("grid", gx, gy, backend="loop")
        st = ko.KrigingState(ndim=2, coords_orig=np.stack([x, y], 1), values=v, model="exponential",
                             params=ko.internal_parameters("exponential", [1.0, 0.3, 0.01]))
        zr, sr = ko.execute(st, "grid", gx, gy)
        good = bool(np.allclose(z, zr, atol=1e-12) and np.allclose(ss, sr, atol=1e-12))
        flow = hdl.calls.count("factor") == (1 if rank == 0 else 0) and hdl.calls.count("bcast_factor") == 1
        # local gather: only this rank's slab comes back
        zl, sl, (lo, hi) = ShardedExecutor(ok, group=pg, handle_factory=lambda: hdl, gather="local").execute("grid", gx, gy, backend="loop")
        local = bool(np.allclose(zl, zr.ravel()[lo:hi], atol=1e-12) and zl.size == hi - lo)
        # rank 0's factorisation fails: every rank raises the same error, nobody waits inside the collective
        bad = RcclStandInHandle(rank, fail_factor=True)
        exb = ShardedExecutor(ok, group=pg, handle_factory=lambda: bad)
        raised = False
        try:
            exb.execute("grid", gx, gy, backend="loop")
        except np.linalg.LinAlgError as e:
            raised = "singular" in str(e)
        # a broadcast that delivers a damaged copy to rank 1: the checksums disagree, EVERY rank factors for itself
        dmg = RcclStandInHandle(rank)
        dmg.corrupt_rank = 1
        exd = ShardedExecutor(ok, group=pg, handle_factory=lambda: dmg)
        zd, ssd = exd.execute("grid", gx, gy, backend="loop")
        damaged = bool(exd.exchange.startswith("redundant_factor (rccl broadcast failed: checksum") and dmg.calls.count("factor") >= 1
                       and np.allclose(zd, zr, atol=1e-12) and "set_grid" in dmg.calls)
        ex.close()
        q.put((rank, good, flow, local, raised and bad.calls.count("bcast_factor") == 0, damaged))
    finally:
        pg.close()


def test_rccl_broadcast_control_flow_world2():
    import multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rccl_flow_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert [o[0] for o in out] == [0, 1] and all(all(o[1:]) for o in out), out


def test_socket_frames_carry_no_pickle_and_strangers_are_dropped():
    import threading

    from pykrige_amd import dist as d

    bufs = []
    tree = d._encode({"a": (1, 2.5, None, b"xy", np.arange(6, dtype=np.int32).reshape(2, 3)), "b": [True, "s"]}, bufs)
    back = d._decode(json_roundtrip(tree), bufs)
    assert back["a"][:3] == (1, 2.5, None) and back["a"][3] == b"xy" and back["b"] == [True, "s"]
    assert back["a"][4].dtype == np.int32 and back["a"][4].tolist() == [[0, 1, 2], [3, 4, 5]]
    with pytest.raises(TypeError):
        d._encode(object(), [])
    with pytest.raises(TypeError):
        d._encode(np.array([object()]), [])
    with pytest.raises(RuntimeError):
        d._decode({"__a": ["|O", [1], 0]}, [b"12345678"])
    # rendezvous: a stranger with the wrong token, then a peer claiming rank 5 of 2, are dropped; the real peer gets in
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    box = {}

    def root():
        box["pg"] = d.SocketGroup(rank=0, world=2, addr="127.0.0.1", port=port, token="secret", timeout=60)

    th = threading.Thread(target=root)
    th.start()
    import struct
    import time

    def knock(rank, token):
        for _ in range(100):
            try:
                c = socket.create_connection(("127.0.0.1", port), timeout=2.0)
                c.sendall(struct.pack("<i", rank) + token.encode().ljust(64, b"\0"))
                return c
            except OSError:
                time.sleep(0.05)
        raise AssertionError("rank 0 never listened")

    c1 = knock(1, "wrong")
    c2 = knock(5, "secret")
    peer = d.SocketGroup(rank=1, world=2, addr="127.0.0.1", port=port, token="secret", timeout=60)
    th.join(60)
    assert not th.is_alive()
    t2 = threading.Thread(target=lambda: box.__setitem__("g0", box["pg"].all_gather_object("r0")))
    t2.start()
    assert peer.all_gather_object("r1") == ["r0", "r1"]
    t2.join(30)
    assert box["g0"] == ["r0", "r1"]
    for c in (c1, c2):
        c.close()
    peer.close()
    box["pg"].close()


def json_roundtrip(tree):
    import json

    return json.loads(json.dumps(tree))
