"""Multi-GPU execute(): one process per GPU, grid points sharded, no cross-GPU reduction.

The only exchange on the data path is the broadcast of the inverted kriging matrix (and c = A_inv.Z)
from rank 0, done by the library over RCCL/xGMI (mik_bcast_factor).  Host-side coordination (the
128-byte RCCL unique id, the final gather of the per-rank slabs) goes through whatever
torch.distributed process group the launcher created -- gloo is enough; torch never touches the device
data.  If RCCL cannot be initialised every rank factors the (identical) matrix itself.

    torchrun --nproc-per-node 8 script.py      # script: dist.init_process_group("gloo"); ShardedExecutor(ok).execute(...)
"""
import os
import pickle
import socket
import struct
import time

import numpy as np


class SocketGroup:
    """Minimal host-side process group over TCP (star through rank 0), enough for what the multi-GPU path needs on the
    host: broadcast of the 128-byte RCCL id, gathers of small Python objects, barriers.  It lets the launcher-provided
    environment (RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT of `python -m torch.distributed.run`) be used WITHOUT importing
    torch, so that the process holds one HIP runtime and one RCCL (the ROCm install's).  Device data never goes through it."""

    def __init__(self, rank=None, world=None, addr=None, port=None, timeout=120.0):
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else int(rank)
        self.world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else int(world)
        addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
        base = int(port if port is not None else int(os.environ.get("MASTER_PORT", "29500")) + 1017)
        if base + 16 > 65535:  # stay inside the port range whatever MASTER_PORT is
            base -= 2 * 1017 + 16
        self._peers = []
        self._sock = None
        if self.world == 1:
            return
        deadline = time.time() + timeout
        if self.rank == 0:
            srv = None
            for off in range(16):  # the launcher's own store sits on MASTER_PORT; take the first free port nearby
                try:
                    srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                    srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                    srv.bind((addr if addr not in ("localhost",) else "127.0.0.1", base + off))
                    break
                except OSError:
                    srv.close()
                    srv = None
            if srv is None:
                raise RuntimeError("SocketGroup: no free port near %d" % base)
            srv.listen(self.world)
            srv.settimeout(max(1.0, deadline - time.time()))
            peers = {}
            while len(peers) < self.world - 1:
                c, _ = srv.accept()
                c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                r = struct.unpack("<i", self._recvn(c, 4))[0]
                peers[r] = c
            srv.close()
            self._peers = [peers[r] for r in range(1, self.world)]
        else:
            last = None
            while time.time() < deadline and self._sock is None:
                for off in range(16):
                    try:
                        c = socket.create_connection((addr, base + off), timeout=2.0)
                        c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                        c.settimeout(None)
                        c.sendall(struct.pack("<i", self.rank))
                        self._sock = c
                        break
                    except OSError as e:
                        last = e
                if self._sock is None:
                    time.sleep(0.2)
            if self._sock is None:
                raise RuntimeError("SocketGroup: cannot reach rank 0 at %s:%d (%r)" % (addr, base, last))

    @staticmethod
    def _recvn(c, n):
        buf = b""
        while len(buf) < n:
            chunk = c.recv(n - len(buf))
            if not chunk:
                raise RuntimeError("SocketGroup: peer closed the connection")
            buf += chunk
        return buf

    def _send(self, c, obj):
        data = pickle.dumps(obj)
        c.sendall(struct.pack("<q", len(data)) + data)

    def _recv(self, c):
        n = struct.unpack("<q", self._recvn(c, 8))[0]
        return pickle.loads(self._recvn(c, n))

    def all_gather_object(self, obj):
        if self.world == 1:
            return [obj]
        if self.rank == 0:
            out = [obj] + [self._recv(c) for c in self._peers]
            for c in self._peers:
                self._send(c, out)
            return out
        self._send(self._sock, obj)
        return self._recv(self._sock)

    def broadcast_object(self, obj, src=0):
        return self.all_gather_object(obj)[src]

    def barrier(self):
        self.all_gather_object(None)

    def all_reduce_max(self, x):
        return max(self.all_gather_object(float(x)))

    def close(self):
        for c in self._peers + ([self._sock] if self._sock else []):
            try:
                c.close()
            except OSError:
                pass
        self._peers, self._sock = [], None


class _TorchGroup:
    """Adapter giving a torch.distributed process group the SocketGroup interface."""

    def __init__(self, group=None):
        import torch.distributed as dist

        if not dist.is_initialized():
            raise RuntimeError("torch.distributed process group is not initialised")
        self._dist, self._group = dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)

    def all_gather_object(self, obj):
        out = [None] * self.world
        self._dist.all_gather_object(out, obj, group=self._group)
        return out

    def broadcast_object(self, obj, src=0):
        box = [obj]
        self._dist.broadcast_object_list(box, src=src, group=self._group)
        return box[0]

    def barrier(self):
        self._dist.barrier(group=self._group)


def slab_bounds(n, world, rank):
    """Contiguous split of n items over `world` ranks (the first n % world ranks get one extra)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    base, extra = divmod(int(n), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


class ShardedExecutor:
    """execute() of a pykrige_amd kriging object with the points sharded over the ranks of a process group
    (a `SocketGroup`, or a torch.distributed group / None = torch's default group).  Every rank must call execute()
    with the same arguments; every rank gets the full result.

    use_rccl=True: rank 0 factors the kriging matrix and the library broadcasts it over RCCL/xGMI (the other ranks do not
    repeat the O(M^3) work but wait for it); use_rccl=False: every rank factors for itself and there is no collective on
    the data path -- never slower in wall-clock on identical GPUs (bench.py measures both and picks)."""

    def __init__(self, model, group=None, use_rccl=True, handle_factory=None):
        self.pg = group if isinstance(group, SocketGroup) else _TorchGroup(group)
        self.model = model
        self.rank, self.world = self.pg.rank, self.pg.world
        self.exchange = "none" if self.world == 1 else ("rccl_bcast" if use_rccl else "redundant_factor")
        self._handle = handle_factory() if handle_factory is not None else model._get_handle()
        if self.exchange == "rccl_bcast":
            self.exchange = init_rccl(self._handle, self.pg)

    def execute(self, style, *axes, mask=None, backend="vectorized", **kw):
        m, h = self.model, self._handle
        window = kw.pop("n_closest_points", None)
        m._check_backend(backend, window)
        pts_adj, shape, fmask, extra = m._prepare_points(style, axes, mask, **kw) if kw else m._prepare_points(style, axes, mask)
        npt = pts_adj.shape[0]
        lo, hi = slab_bounds(npt, self.world, self.rank)
        m._set_problem(h)
        if window is None:
            if self.exchange == "rccl_bcast":
                if self.rank == 0:
                    h.factor()
                h.bcast_factor(0)
            else:
                h.factor()
        sl = slice(lo, hi)
        h.set_points(pts_adj[sl, 0], pts_adj[sl, 1], pts_adj[sl, 2] if m._ndim == 3 else None,
                     mask=None if fmask is None else fmask[sl], extra_rows=None if extra is None else extra[:, sl])
        if window is None:
            h.predict()
        else:  # moving window: every point needs only its own neighbours -- no factor, no exchange of any kind
            h.predict_moving_window(int(window))
        z, ss = h.get_results()
        parts = self.pg.all_gather_object((lo, z, ss))
        zf, sf = np.zeros(npt), np.zeros(npt)
        for plo, pz, ps in parts:
            zf[plo:plo + pz.size] = pz
            sf[plo:plo + ps.size] = ps
        return m._finish(zf, sf, style, shape, fmask, backend)


def init_rccl(handle, pg, timeout=None):
    """Create the library's RCCL communicator on every rank of `pg`.  Returns "rccl_bcast" when every rank joined, else a
    "redundant_factor (...)" string (then every rank factors the matrix itself).  ncclCommInitRank blocks until all ranks
    have joined, so it runs in a thread: a wedged bootstrap degrades instead of hanging the job."""
    import threading

    from . import _lib

    timeout = float(os.environ.get("MIK_RCCL_INIT_TIMEOUT", "120")) if timeout is None else timeout
    err = None
    try:
        uid = _lib.Handle.comm_unique_id() if pg.rank == 0 else None
    except Exception as e:  # rank 0 cannot even load RCCL: tell everybody
        uid, err = None, repr(e)[:120]
    uid = pg.broadcast_object(uid, src=0)
    if uid is not None:
        box = {}

        def _init():
            try:
                handle.comm_init(pg.world, pg.rank, uid)
                box["ok"] = True
            except Exception as e:  # noqa: BLE001
                box["err"] = repr(e)[:120]

        th = threading.Thread(target=_init, daemon=True)
        th.start()
        th.join(timeout)
        if not box.get("ok"):
            err = box.get("err", "ncclCommInitRank timed out after %.0f s" % timeout)
    else:
        err = err or "rank 0 could not create an RCCL unique id"
    errs = pg.all_gather_object(err)
    bad = [e for e in errs if e is not None]
    return "rccl_bcast" if not bad else "redundant_factor (rccl unavailable: %s)" % bad[0]
