"""Multi-GPU execute() with ONE PROCESS PER GPU (`python -m torch.distributed.run --nproc-per-node 8 script.py`): grid
points sharded over the ranks, no cross-GPU reduction.

(On a single node nothing of this is needed: `pykrige_amd.set_devices(8)` / MIK_NGPU=8 makes every kriging object's
handle span the GPUs inside one process -- include/mikrige.h, mik_set_devices.  This module is the launcher-based form.)

The only exchange on the data path is the broadcast of the inverted kriging matrix (and c = A_inv.Z) from rank 0, done by
the library over RCCL/xGMI (mik_bcast_factor).  Host-side coordination -- the 128-byte RCCL unique id, status agreement,
barriers -- goes through `SocketGroup`, a small TCP star built from the launcher's RANK / WORLD_SIZE / MASTER_ADDR /
MASTER_PORT, so the process never imports torch and holds one HIP runtime and one RCCL (a torch.distributed group is
accepted too).  The per-rank slabs of z / sigma^2 meet in a shared-memory segment (ranks of one node), not in a socket.
If RCCL cannot be initialised every rank factors the (identical) matrix itself.
"""
import hmac
import json
import os
import socket
import struct
import time

import numpy as np

_ARRAY_KINDS = "biuf"  # dtypes a peer may send: bool, signed / unsigned int, float -- never objects


def _encode(obj, bufs):
    """Python value -> JSON-able tree; ndarrays / bytes travel as raw buffers behind the header.  No pickle: nothing a
    peer sends is ever executed."""
    if obj is None or isinstance(obj, (bool, int, float, str)):
        return obj
    if isinstance(obj, (np.integer,)):
        return int(obj)
    if isinstance(obj, (np.floating,)):
        return float(obj)
    if isinstance(obj, (bytes, bytearray)):
        bufs.append(bytes(obj))
        return {"__b": len(bufs) - 1}
    if isinstance(obj, np.ndarray):
        if obj.dtype.kind not in _ARRAY_KINDS:
            raise TypeError("SocketGroup: cannot send arrays of dtype %s" % obj.dtype)
        a = np.ascontiguousarray(obj)
        bufs.append(a.tobytes())
        return {"__a": [a.dtype.str, list(a.shape), len(bufs) - 1]}
    if isinstance(obj, tuple):
        return {"__t": [_encode(v, bufs) for v in obj]}
    if isinstance(obj, list):
        return [_encode(v, bufs) for v in obj]
    if isinstance(obj, dict):
        return {"__d": [[_encode(k, bufs), _encode(v, bufs)] for k, v in obj.items()]}
    raise TypeError("SocketGroup: cannot send a %s" % type(obj).__name__)


def _decode(node, bufs):
    if isinstance(node, list):
        return [_decode(v, bufs) for v in node]
    if isinstance(node, dict):
        if "__b" in node:
            return bufs[int(node["__b"])]
        if "__a" in node:
            dt, shape, i = node["__a"]
            dtype = np.dtype(str(dt))
            if dtype.kind not in _ARRAY_KINDS:
                raise RuntimeError("SocketGroup: peer sent an array of dtype %s" % dt)
            return np.frombuffer(bufs[int(i)], dtype=dtype).reshape([int(s) for s in shape]).copy()
        if "__t" in node:
            return tuple(_decode(v, bufs) for v in node["__t"])
        if "__d" in node:
            return {_decode(k, bufs): _decode(v, bufs) for k, v in node["__d"]}
        raise RuntimeError("SocketGroup: malformed frame")
    return node


class SocketGroup:
    """Minimal host-side process group over TCP (star through rank 0), enough for what the multi-GPU path needs on the
    host: broadcast of the 128-byte RCCL id, gathers of small values, barriers.  It lets the launcher-provided
    environment (RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT of `python -m torch.distributed.run`) be used WITHOUT importing
    torch, so that the process holds one HIP runtime and one RCCL (the ROCm install's).  Device data never goes through it.

    Frames are length-prefixed JSON headers followed by raw buffers (no pickle).  A joining peer must present the job's
    token (MIK_SOCKET_TOKEN, else the launcher's TORCHELASTIC_RUN_ID) and a rank in 1..world-1 that nobody else has claimed;
    anything else is dropped (constant-time comparison).  Without a token the group only forms on the loopback interface:
    a port any host can reach must not hand out ranks to whoever connects first.  A frame is capped at `max_frame` bytes
    (default 256 MiB: the group carries ids, statuses and -- multi-node gathers only -- result slabs; callers that expect
    more raise it for that call), so an accepted peer cannot make rank 0 allocate without bound."""

    MAX_FRAME = 1 << 28

    def __init__(self, rank=None, world=None, addr=None, port=None, timeout=120.0, token=None):
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else int(rank)
        self.world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else int(world)
        addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
        base = int(port if port is not None else int(os.environ.get("MASTER_PORT", "29500")) + 1017)
        if base + 16 > 65535:  # stay inside the port range whatever MASTER_PORT is
            base -= 2 * 1017 + 16
        if token is None:
            token = os.environ.get("MIK_SOCKET_TOKEN", os.environ.get("TORCHELASTIC_RUN_ID", ""))
        self.max_frame = self.MAX_FRAME
        if not str(token) and self.world > 1 and addr not in ("127.0.0.1", "localhost", "::1"):
            raise RuntimeError("SocketGroup: refusing to form a group on %s without a job token (set MIK_SOCKET_TOKEN, or launch "
                               "through torch.distributed.run which provides TORCHELASTIC_RUN_ID)" % addr)
        self._token = str(token).encode("utf-8")[:64].ljust(64, b"\0")
        self._peers = []
        self._sock = None
        if not (0 <= self.rank < self.world):
            raise ValueError("SocketGroup: rank %d outside world %d" % (self.rank, self.world))
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
            peers = {}
            while len(peers) < self.world - 1:
                srv.settimeout(max(1.0, deadline - time.time()))
                c, _ = srv.accept()
                try:
                    c.settimeout(10.0)
                    hello = self._recvn(c, 68)
                    r = struct.unpack("<i", hello[:4])[0]
                    if not hmac.compare_digest(hello[4:], self._token) or not (1 <= r < self.world) or r in peers:
                        raise RuntimeError("bad hello")
                    c.settimeout(None)
                    c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    peers[r] = c
                except Exception:  # noqa: BLE001  -- a stranger, a duplicate or a wrong token: drop it, keep listening
                    c.close()
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
                        c.sendall(struct.pack("<i", self.rank) + self._token)
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
        buf = bytearray()
        while len(buf) < n:
            chunk = c.recv(min(n - len(buf), 1 << 20))
            if not chunk:
                raise RuntimeError("SocketGroup: peer closed the connection")
            buf += chunk
        return bytes(buf)

    def _send(self, c, obj):
        bufs = []
        head = json.dumps({"v": _encode(obj, bufs), "bufs": [len(b) for b in bufs]}).encode("utf-8")
        c.sendall(struct.pack("<q", len(head)) + head + b"".join(bufs))

    def _recv(self, c):
        n = struct.unpack("<q", self._recvn(c, 8))[0]
        if not (0 < n <= self.max_frame):
            raise RuntimeError("SocketGroup: malformed frame length")
        head = json.loads(self._recvn(c, n).decode("utf-8"))
        sizes = [int(s) for s in head["bufs"]]
        if any(s < 0 for s in sizes) or sum(sizes) > self.max_frame:
            raise RuntimeError("SocketGroup: malformed buffer table")
        bufs = [self._recvn(c, s) for s in sizes]
        return _decode(head["v"], bufs)

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


class _SharedResults:
    """z and sigma^2 of the whole grid in ONE shared-memory segment of the node: rank 0 creates it, every rank writes its
    slab in place, a barrier, everybody reads.  Replaces a gather of pickled arrays through rank 0's sockets (268 MB in and
    7 x 268 MB out per call at config 5)."""

    def __init__(self, pg):
        self.pg, self.shm, self.npt = pg, None, -1

    def arrays(self, npt):
        from multiprocessing import shared_memory

        if npt != self.npt:
            self.release()
            name = None
            if self.pg.rank == 0:
                self.shm = shared_memory.SharedMemory(create=True, size=max(16, 16 * npt))
                name = self.shm.name
            name = self.pg.broadcast_object(name, src=0)
            if self.pg.rank != 0:
                self.shm = shared_memory.SharedMemory(name=name)
                try:  # the creator unlinks; keep Python's resource tracker of the other ranks from doing it a second time
                    from multiprocessing import resource_tracker

                    resource_tracker.unregister(self.shm._name, "shared_memory")
                except Exception:  # noqa: BLE001
                    pass
            self.npt = npt
        both = np.ndarray((2, npt), dtype=np.float64, buffer=self.shm.buf)
        return both[0], both[1]

    def release(self):
        if self.shm is not None:
            self.pg.barrier()  # nobody still reads
            self.shm.close()
            if self.pg.rank == 0:
                self.shm.unlink()
            self.shm, self.npt = None, -1


def _ranks_share_a_node(pg):
    """True when every rank can open a shared-memory segment rank 0 created and reads rank 0's nonce in it.  (Comparing host
    names is not enough: cloned images and containers named alike share a name without sharing /dev/shm; results would then
    silently be each node's own slabs.)"""
    from multiprocessing import shared_memory

    shm, name, nonce = None, None, None
    if pg.rank == 0:
        try:
            nonce = os.urandom(16)
            shm = shared_memory.SharedMemory(create=True, size=16)
            shm.buf[:16] = nonce
            name = shm.name
        except Exception:  # noqa: BLE001
            shm, name = None, None
    name, nonce = pg.broadcast_object((name, nonce), src=0)
    ok = name is not None
    if ok and pg.rank != 0:
        try:
            other = shared_memory.SharedMemory(name=name)
            try:
                from multiprocessing import resource_tracker

                resource_tracker.unregister(other._name, "shared_memory")  # the creator unlinks
            except Exception:  # noqa: BLE001
                pass
            ok = bytes(other.buf[:16]) == nonce
            other.close()
        except Exception:  # noqa: BLE001
            ok = False
    oks = pg.all_gather_object(bool(ok))  # (also: nobody still has the probe open when rank 0 unlinks it)
    if shm is not None:
        shm.close()
        shm.unlink()
    return all(oks)


class ShardedExecutor:
    """execute() of a pykrige_amd kriging object with the points sharded over the ranks of a process group: group=None
    builds a `SocketGroup` from the launcher's environment (no torch); a `SocketGroup` or a torch.distributed group can be
    passed.  Every rank must call execute() with the same arguments.

    gather="shared" (default): every rank gets the full result; the slabs meet in a shared-memory segment (the ranks share
    a node).  gather="local": every rank gets only its own slab -- `(z, ss, (lo, hi))`, flat, in the reference's point
    order -- for callers that write their part of the output themselves (or span several nodes).

    use_rccl=True: rank 0 factors the kriging matrix and the library broadcasts it over RCCL/xGMI (the other ranks do not
    repeat the O(M^3) work but wait for it); use_rccl=False: every rank factors for itself and there is no collective on
    the data path."""

    def __init__(self, model, group=None, use_rccl=True, handle_factory=None, gather="shared"):
        if group is None:
            self.pg = SocketGroup()
        elif isinstance(group, SocketGroup) or (hasattr(group, "all_gather_object") and hasattr(group, "world")):
            self.pg = group
        else:
            self.pg = _TorchGroup(group)
        if gather not in ("shared", "local"):
            raise ValueError("gather must be 'shared' or 'local'")
        self.gather = gather
        self.model = model
        self.rank, self.world = self.pg.rank, self.pg.world
        self.exchange = "none" if self.world == 1 else ("rccl_bcast" if use_rccl else "redundant_factor")
        self.exchange_ms = 0.0
        self._handle = handle_factory() if handle_factory is not None else model._get_handle()
        if self.exchange == "rccl_bcast":
            self.exchange = init_rccl(self._handle, self.pg)
        # a shared-memory segment only exists on ONE node: ranks on different hosts gather through the group instead
        self.single_node = _ranks_share_a_node(self.pg) if self.world > 1 else True
        self._shared = _SharedResults(self.pg)

    def close(self):
        self._shared.release()

    def _factor_everywhere(self, h):
        """Rank 0 factors, everybody learns how that went BEFORE anybody enters the collective (a rank waiting inside
        ncclBroadcast for a root that raised would wait forever); then one broadcast.  A failed factorisation raises the
        same exception on every rank."""
        status = None
        if self.rank == 0:
            try:
                h.factor()
            except Exception as e:  # noqa: BLE001
                status = (type(e).__name__, str(e))
        status = self.pg.broadcast_object(status, src=0)
        if status is not None:
            kind, msg = status
            exc = {"LinAlgError": np.linalg.LinAlgError, "ValueError": ValueError}.get(kind, RuntimeError)
            raise exc(msg if self.rank == 0 else "rank 0: " + msg)
        err, sums = None, None
        try:
            t0 = time.perf_counter()
            h.bcast_factor(0)  # bounded inside the library (MIK_RCCL_BCAST_TIMEOUT): returns an error instead of hanging
            self.exchange_ms = (time.perf_counter() - t0) * 1e3  # this rank's wall time inside the broadcast of the packed upper triangle + c
            sums = h.factor_checksum() if hasattr(h, "factor_checksum") else None
        except Exception as e:  # noqa: BLE001
            err = repr(e)[:200]
        res = self.pg.all_gather_object((err, sums))
        errs = [e for e, _ in res if e]
        if not errs and any(sm != res[0][1] for _, sm in res):  # a copy of the inverse that differs from the root's
            errs = ["checksum of the broadcast inverse differs from rank 0's on rank(s) %s"
                    % [r for r, (_, sm) in enumerate(res) if sm != res[0][1]]]
        if errs:  # the collective itself failed somewhere: from now on (and for this call) every rank factors itself
            self.exchange = "redundant_factor (rccl broadcast failed: %s)" % errs[0]
            h.factor()

    def execute(self, style, *axes, mask=None, backend="vectorized", **kw):
        m, h = self.model, self._handle
        window = kw.pop("n_closest_points", None)
        m._check_backend(backend, window)
        P = m._prepare(style, axes, mask, kw.get("specified_drift_arrays"), backend)  # a grid stays axes: no host meshgrid
        npt, shape, fmask = P.npt, P.shape, P.mask
        lo, hi = slab_bounds(npt, self.world, self.rank)
        m._set_problem(h)
        if window is None:
            if self.exchange == "rccl_bcast":
                self._factor_everywhere(h)
            else:
                h.factor()
        P.load(h, m._ndim, cell_range=(lo, hi - lo), with_extra=window is None)
        if window is None:
            h.predict()
        else:  # moving window: every point needs only its own neighbours -- no factor, no exchange of any kind
            h.predict_moving_window(int(window))
        z, ss = h.get_results()
        if self.gather == "local":
            return z, ss, (lo, hi)
        if self.world == 1:
            zf, sf = z, ss
        elif not self.single_node:  # ranks on several hosts: the slabs travel through the group (raw buffers, no pickle)
            if hasattr(self.pg, "max_frame"):
                self.pg.max_frame = max(self.pg.max_frame, 32 * npt + (1 << 20))
            parts = self.pg.all_gather_object((z, ss))
            zf, sf = np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts])
        else:
            zs, sshared = self._shared.arrays(npt)
            zs[lo:hi], sshared[lo:hi] = z, ss
            self.pg.barrier()  # every slab is in place
            zf, sf = zs.copy(), sshared.copy()
            self.pg.barrier()  # every rank has read: the segment may be reused by the next call
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
