"""TEST INFRASTRUCTURE (checker only; pykrige_amd never imports it).  Parity over WHOLE grids against the REAL reference.

The stored slabs under tests/golden/fullsize cover 0.1 - 1.7 % of the grids BASELINE.json names.  Here the staged reference
(oracle/ref_package.import_reference(): ok.py / uk.py / ok3d.py as written upstream) kriges the grid the claim is made about,
`execute('grid', slab axes, backend='vectorized')` slab by slab -- a slab being whole rows (2-D) or whole z planes (3-D) of the
config's own grid, sized so that the reference's npt x N temporaries (bd, b, x: ok.py:650-683) stay within a few GB -- and every
point of one `execute('grid')` of the drop-in class is compared with it.  The reference assembles and inverts the kriging matrix
once per call (ok.py:898, 663); nothing of it is patched or memoised.

A wall-clock budget can bound the CPU side (the -m gpu suite must finish under the driver's limit on any box): slabs are visited
in van der Corput order, so whatever fraction fits is spread over the whole grid, and the fraction is reported.
"""
import time

import numpy as np


def spread_order(n):
    """0 .. n-1 in bit-reversal (van der Corput) order: any prefix of it is spread evenly over the range."""
    bits = max(1, (max(n, 1) - 1).bit_length())
    return sorted(range(n), key=lambda i: int(format(i, "0%db" % bits)[::-1], 2))


def slab_plan(axes, target_points):
    """[(slab axes, slice of the slowest axis)]: whole rows (2-D: result[y, x]) or whole z planes (3-D: result[z, y, x]) of the grid,
    about `target_points` points each (at least one row / plane), in grid order."""
    slow = axes[-1]
    per = int(np.prod([a.size for a in axes[:-1]]))
    step = max(1, int(target_points) // per)
    return [(list(axes[:-1]) + [slow[s:s + step]], slice(s, min(s + step, slow.size))) for s in range(0, slow.size, step)]


# ---- several reference processes side by side.  The reference's _exec_vector is mostly SINGLE-THREADED NumPy / SciPy (cdist, the variogram's exp
# over npt x N, three passes over the npt x N products: ok.py:669-681); only its np.dot uses the BLAS threads.  One process therefore leaves a 64-core
# host idle most of the time (5.6 k points/s at config 2); W processes with cores / W BLAS threads each krige W slabs at once.  Every process builds
# the reference model from the same recipe and runs the reference's own execute() unmodified.
_WORKER = {}


def _worker_init(recipe, threads):
    import os
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    try:
        import threadpoolctl

        _WORKER["limit"] = threadpoolctl.threadpool_limits(limits=max(1, int(threads)))
    except Exception:  # noqa: BLE001
        pass
    import bench
    from oracle import ref_package as rp

    cfg, coords, values = recipe
    _WORKER["model"] = bench.reference_model(rp.import_reference(stub_statistics=True), cfg, coords, values)


def _worker_slab(job):
    i, slab_axes, backend = job
    t1 = time.perf_counter()
    zr, sr = _WORKER["model"].execute("grid", *slab_axes, backend=backend)
    return i, np.ma.getdata(zr), np.ma.getdata(sr), time.perf_counter() - t1


def _make_pool(recipe, workers):
    import multiprocessing as mp
    import os

    # 8 BLAS threads per process, set through the ENVIRONMENT the children are born with: OpenBLAS then creates 8 threads, not 64 that spin between
    # calls (8 processes x 32 threads limited after the fact took 1 000 s per slab instead of 9 on the 256-CPU box: profiles/r06_full_grid_parity.txt)
    threads = max(1, min(8, (os.cpu_count() or 1) // workers))
    saved = {k: os.environ.get(k) for k in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS")}
    for k in saved:
        os.environ[k] = str(threads)
    ctx = mp.get_context("spawn")  # (never fork a process that holds a HIP runtime)
    try:
        return ctx.Pool(workers, initializer=_worker_init, initargs=(recipe, threads))
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _slab_results(ref_model, plan, order, backend, budget_s, recipe, workers):
    """(i, zr, sr, seconds) of the slabs in `order`, as far as `budget_s` reaches (whole slabs, at least one)."""
    t0 = time.perf_counter()
    pool = None
    if recipe and workers > 1 and len(order) >= 2:
        try:
            pool = _make_pool(recipe, min(workers, len(order)))
        except Exception:  # noqa: BLE001  (no processes to be had: one process, as before)
            pool = None
    if pool is None:
        last, n = 0.0, 0
        for i in order:
            if budget_s is not None and n and time.perf_counter() - t0 + last > budget_s:
                return
            t1 = time.perf_counter()
            zr, sr = ref_model.execute("grid", *plan[i][0], backend=backend)
            last = time.perf_counter() - t1
            n += 1
            yield i, np.ma.getdata(zr), np.ma.getdata(sr), last
        return
    try:
        pending, it, longest = [], iter(order), 0.0
        for _ in range(workers):
            i = next(it, None)
            if i is not None:
                pending.append(pool.apply_async(_worker_slab, ((i, plan[i][0], backend),)))
        done = 0
        while pending:
            res = pending.pop(0).get(timeout=1800)  # (a worker that cannot start never answers: an error, not a hang)
            longest = max(longest, res[3])
            done += 1
            yield res
            # a new slab only if it can be expected to finish inside the budget (slabs take about as long as the longest one seen)
            if budget_s is None or time.perf_counter() - t0 + longest <= budget_s:
                i = next(it, None)
                if i is not None:
                    pending.append(pool.apply_async(_worker_slab, ((i, plan[i][0], backend),)))
    finally:
        pool.terminate()
        pool.join()


def compare(ref_model, z_gpu, ss_gpu, axes, target_points, budget_s=None, backend="vectorized", log=None, recipe=None, workers=1):
    """Krige the grid `axes` with the reference model slab by slab and compare with the drop-in's whole-grid result (arrays shaped
    as the reference returns them: [y, x] or [z, y, x]).  Returns a dict of scalars; stops early (whole slabs only, at least one)
    when the next slab would overrun `budget_s`.  recipe = (cfg, coords, values) + workers > 1: that many reference processes side by side
    (each builds bench.reference_model(cfg, coords, values) and runs its execute() as written)."""
    plan = slab_plan(axes, target_points)
    z_gpu, ss_gpu = np.ma.getdata(z_gpu), np.ma.getdata(ss_gpu)
    total = int(np.prod([a.size for a in axes]))
    assert z_gpu.shape == tuple(a.size for a in reversed(axes)), (z_gpu.shape, [a.size for a in axes])
    out = {"points_total": total, "points_checked": 0, "slabs_total": len(plan), "slabs_checked": 0, "max_abs_dz": 0.0, "max_abs_dss": 0.0,
           "worst_dz_at": None, "worst_dss_at": None, "backend": backend, "reference_processes": max(1, workers if recipe else 1)}
    t0 = time.perf_counter()
    for i, zr, sr, last in _slab_results(ref_model, plan, spread_order(len(plan)), backend, budget_s, recipe, workers):
        slab_axes, sl = plan[i]
        dz, ds = np.abs(z_gpu[sl] - zr), np.abs(ss_gpu[sl] - sr)
        if not (np.isfinite(dz).all() and np.isfinite(ds).all()):
            dz, ds = np.where(np.isfinite(dz), dz, np.inf), np.where(np.isfinite(ds), ds, np.inf)
        for key, d, at in (("max_abs_dz", dz, "worst_dz_at"), ("max_abs_dss", ds, "worst_dss_at")):
            m = float(d.max())
            if m >= out[key]:
                j = np.unravel_index(int(np.argmax(d)), d.shape)  # (slow .. fast) index inside the slab
                out[key] = m
                out[at] = [float(a[k]) for a, k in zip(slab_axes, reversed(j))]
        out["points_checked"] += int(zr.size)
        out["slabs_checked"] += 1
        if log:
            log("  slab %d/%d (%s %d:%d): %d points in %.1f s, running max|dz| %.2e max|dss| %.2e" % (
                out["slabs_checked"], len(plan), "rows" if len(axes) == 2 else "planes", sl.start, sl.stop, zr.size, last,
                out["max_abs_dz"], out["max_abs_dss"]))
    out["reference_s"] = time.perf_counter() - t0
    out["reference_points_per_s"] = out["points_checked"] / max(out["reference_s"], 1e-9)
    out["coverage"] = out["points_checked"] / float(total)
    return out


def cond_1(ref_model, *matrix_args):
    """1-norm condition number of the REFERENCE's kriging matrix (the context of the tolerance): ||A||_1 ||A^-1||_1."""
    import scipy.linalg

    a = np.array(ref_model._get_kriging_matrix(*matrix_args))
    na = np.abs(a).sum(axis=0).max()
    ai = scipy.linalg.inv(a, overwrite_a=True)
    return float(na * np.abs(ai).sum(axis=0).max())
