"""Multi-GPU execute(): one process per GPU, grid points sharded, no cross-GPU reduction.

The only exchange on the data path is the broadcast of the inverted kriging matrix (and c = A_inv.Z)
from rank 0, done by the library over RCCL/xGMI (mik_bcast_factor).  Host-side coordination (the
128-byte RCCL unique id, the final gather of the per-rank slabs) goes through whatever
torch.distributed process group the launcher created -- gloo is enough; torch never touches the device
data.  If RCCL cannot be initialised every rank factors the (identical) matrix itself.

    torchrun --nproc-per-node 8 script.py      # script: dist.init_process_group("gloo"); ShardedExecutor(ok).execute(...)
"""
import numpy as np


def slab_bounds(n, world, rank):
    """Contiguous split of n items over `world` ranks (the first n % world ranks get one extra)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    base, extra = divmod(int(n), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


class ShardedExecutor:
    """execute() of a pykrige_amd kriging object with the points sharded over the ranks of a
    torch.distributed process group.  Every rank must call execute() with the same arguments; every
    rank gets the full result."""

    def __init__(self, model, group=None, use_rccl=True, handle_factory=None):
        import torch.distributed as dist

        if not dist.is_initialized():
            raise RuntimeError("torch.distributed process group is not initialised")
        self._dist = dist
        self.model = model
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.exchange = "none" if self.world == 1 else ("rccl_bcast" if use_rccl else "redundant_factor")
        self._handle = handle_factory() if handle_factory is not None else model._get_handle()
        if self.exchange == "rccl_bcast":
            self._init_comm()

    def _init_comm(self):
        from . import _lib

        dist, err = self._dist, None
        try:
            uid = [_lib.Handle.comm_unique_id() if self.rank == 0 else None]
        except Exception as e:  # rank 0 cannot even load RCCL: tell everybody
            uid, err = [None], repr(e)
        dist.broadcast_object_list(uid, src=0, group=self.group)
        if uid[0] is not None:
            try:
                self._handle.comm_init(self.world, self.rank, uid[0])
            except Exception as e:
                err = repr(e)
        else:
            err = err or "rank 0 could not create an RCCL unique id"
        errs = [None] * self.world
        dist.all_gather_object(errs, err, group=self.group)
        if any(e is not None for e in errs):
            self.exchange = "redundant_factor (rccl unavailable: %s)" % next(e for e in errs if e is not None)

    def execute(self, style, *axes, mask=None, backend="vectorized", **kw):
        m, h, dist = self.model, self._handle, self._dist
        m._check_backend(backend, kw.pop("n_closest_points", None))
        pts_adj, shape, fmask, extra = m._prepare_points(style, axes, mask, **kw) if kw else m._prepare_points(style, axes, mask)
        npt = pts_adj.shape[0]
        lo, hi = slab_bounds(npt, self.world, self.rank)
        m._set_problem(h)
        if self.exchange == "rccl_bcast":
            if self.rank == 0:
                h.factor()
            h.bcast_factor(0)
        else:
            h.factor()
        sl = slice(lo, hi)
        h.set_points(pts_adj[sl, 0], pts_adj[sl, 1], pts_adj[sl, 2] if m._ndim == 3 else None,
                     mask=None if fmask is None else fmask[sl], extra_rows=None if extra is None else extra[:, sl])
        h.predict()
        z, ss = h.get_results()
        parts = [None] * self.world
        dist.all_gather_object(parts, (lo, z, ss), group=self.group)
        zf, sf = np.zeros(npt), np.zeros(npt)
        for plo, pz, ps in parts:
            zf[plo:plo + pz.size] = pz
            sf[plo:plo + ps.size] = ps
        return m._finish(zf, sf, style, shape, fmask, backend)
