#!/usr/bin/env python
"""bench.py -- kriged grid-points/sec (z + sigma^2) of the HIP execute() path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 2|3|4|5] [--no-cpu] [--moving-window K]

One "step" = ONE call of the drop-in class's execute('grid', axes) end to end ("wall-clock of execute()"):
host front matter, the grid generated on the device from its axes (mik_set_grid: what crosses PCIe on the way in is
O(nx + ny)), kriging-matrix assembly + inverse (K1, K2) [+ the exchange of the inverse when N > 1, overlapped with the
leader's prediction] + RHS assembly and contraction (K3) for every grid point; z and sigma^2 land in page-locked host
memory chunk by chunk while the next chunk is computed and are copied into the returned arrays.  The rate with the points
already resident in HBM (mik_factor + mik_predict per step) stands beside it as `resident`.
Default workload = BASELINE.json configs[1]: OrdinaryKriging 2D, N=5000 stations, 1000x1000 grid,
exponential variogram [1.0, 0.3, 0.0], fp64, synthetic stations (seed 2).

N > 1 is WEAK scaling: the grid grows to 1000 x (1000 N) (config 5: N slabs of 4096 x 512), every GPU kriges one slab
against the same stations; the leader assembles + inverts and the inverse is broadcast (RCCL over xGMI); no other
collective.  Two launch forms:
  * plain `python bench.py --gpus N`: ONE process, the library's device group (mik_set_devices: one handle spans N GPUs,
    one host thread and one stream per GPU, ncclCommInitAll + grouped ncclBroadcast).  On a box with fewer than N GPUs the
    group aliases devices (several members on one GPU) so that the path can be exercised anywhere; the JSON says so.
  * under `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`: one process per GPU, rank r kriges slab r,
    mik_comm_init + mik_bcast_factor; host rendezvous / barrier / max-over-ranks through pykrige_amd.dist.SocketGroup
    (TCP, from the launcher's RANK/WORLD_SIZE/MASTER_* environment; torch is not imported unless that fails).
Rank 0 prints ONE JSON line.
"""
import argparse
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6  # MI355X FP64 matrix peak (public spec; = 256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz).
# The local microarch guide lists no FP64 row; tools/ubench_f64.hip measures what the part sustains:
FP64_MFMA_MEASURED_TFLOPS = 73.2  # v_mfma_f64_4x4x4_4b_f64, one wave per SIMD issuing back to back (profiles/r01_ubench_f64.txt)

CONFIGS = {
    # name: (ndim, seed, n, axes sizes (x, y[, z]), model, user params, drift)
    2: dict(name="OK2D N=5000 1000x1000 exponential", ndim=2, seed=2, n=5000, grid=(1000, 1000), model="exponential",
            params=[1.0, 0.3, 0.0]),
    3: dict(name="OK3D N=2000 200x200x50 gaussian", ndim=3, seed=3, n=2000, grid=(200, 200, 50), model="gaussian",
            params=[1.0, 0.4, 0.02]),
    4: dict(name="UK2D N=4000 1024x1024 exponential regional_linear+point_log", ndim=2, seed=4, n=4000,
            grid=(1024, 1024), model="exponential", params=[1.0, 0.3, 0.01], rl=True,
            wells=[[0.3137, 0.7219, 1.0], [0.6621, 0.2483, -0.5], [0.8412, 0.8127, 2.0]]),
    5: dict(name="OK2D N=8000 4096x4096 spherical (per-GPU shard = 4096x512 rows)", ndim=2, seed=5, n=8000,
            grid=(4096, 512), model="spherical", params=[1.0, 0.2, 0.01]),
}
EXCHANGE_CODES = {"auto": 0, "rccl": 1, "peer": 2, "redundant": 3}
EXCHANGE_NAMES = {0: "none", 1: "rccl_bcast", 2: "peer_scatter_allgather", 3: "redundant_factor"}


def synth(seed, n, ndim):
    rng = np.random.default_rng(seed)
    c = [rng.random(n) for _ in range(ndim)]
    v = np.sin(6 * c[0]) * np.cos(4 * c[1])
    if ndim == 3:
        v = v * np.cos(3 * c[2])
    return c, v + 0.1 * rng.standard_normal(n)


def internal_params(model, p):
    return [p[0] - p[2], p[1], p[2]] if model in ("gaussian", "spherical", "exponential", "hole-effect") else list(p)


def shard_points(cfg, rank, world):
    """This rank's slab of the (weak-scaled) grid, flattened in the reference's meshgrid order."""
    g = cfg["grid"]
    if cfg["ndim"] == 2:
        nx, ny = g
        gx = np.linspace(0.0, 1.0, nx)
        gy_all = np.linspace(0.0, 1.0, ny * world)
        gy = gy_all[rank * ny:(rank + 1) * ny]
        X, Y = np.meshgrid(gx, gy)
        return [X.ravel(), Y.ravel()]
    nx, ny, nz = g
    gx, gy = np.linspace(0.0, 1.0, nx), np.linspace(0.0, 1.0, ny)
    gz_all = np.linspace(0.0, 1.0, nz * world)
    gz = gz_all[rank * nz:(rank + 1) * nz]
    Z, Y, X = np.meshgrid(gz, gy, gx, indexing="ij")
    return [X.ravel(), Y.ravel(), Z.ravel()]


def row_slab_axes(cfg, min_points):
    """Axes of a row slab of the config's own grid with >= min_points points, whole rows from the middle of it (BASELINE.md
    section 3: the reference is timed on row slabs via style='grid' with a y-subrange)."""
    g = cfg["grid"]
    nx = g[0]
    rows = int(np.ceil(min_points / nx))
    if cfg["ndim"] == 2:
        ny_all = 4096 if cfg["n"] == 8000 else g[1]  # config 5's grid is 4096 x 4096 (CONFIGS holds one GPU's 512 rows)
        return [np.linspace(0.0, 1.0, nx), np.linspace(0.0, 1.0, ny_all)[ny_all // 2:ny_all // 2 + rows]]
    return [np.linspace(0.0, 1.0, nx), np.linspace(0.0, 1.0, g[1])[g[1] // 4:g[1] // 4 + rows],
            np.linspace(0.0, 1.0, g[2])[g[2] // 2:g[2] // 2 + 1]]


def row_slab(cfg, min_points):
    """The points of row_slab_axes in the reference's meshgrid order."""
    ax = row_slab_axes(cfg, min_points)
    if cfg["ndim"] == 2:
        X, Y = np.meshgrid(ax[0], ax[1])
        return np.stack([X.ravel(), Y.ravel()], 1)
    Z, Y, X = np.meshgrid(ax[2], ax[1], ax[0], indexing="ij")
    return np.stack([X.ravel(), Y.ravel(), Z.ravel()], 1)


def reference_model(pk, cfg, coords, values):
    """The REAL reference's class for a config (oracle/ref_package.py: the package as written upstream, staged by
    oracle/build_ref.sh), constructor statistics stubbed as BASELINE.md section 3 prescribes."""
    kw = dict(variogram_model=cfg["model"], variogram_parameters=list(cfg["params"]))
    if cfg["ndim"] == 3:
        return pk.ok3d.OrdinaryKriging3D(coords[0], coords[1], coords[2], values, **kw)
    if cfg.get("rl") or cfg.get("wells"):
        terms = (["regional_linear"] if cfg.get("rl") else []) + (["point_log"] if cfg.get("wells") else [])
        return pk.uk.UniversalKriging(coords[0], coords[1], values, drift_terms=terms, point_drift=cfg.get("wells"), **kw)
    return pk.ok.OrdinaryKriging(coords[0], coords[1], values, **kw)


def host_description():
    """CPU model, logical CPUs, BLAS library / version / threads."""
    d = {"host_cpus": os.cpu_count(), "numpy": np.__version__}
    try:
        import scipy

        d["scipy"] = scipy.__version__
    except Exception:
        pass
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                d["cpu_model"] = line.split(":", 1)[1].strip()
                break
    except Exception:
        pass
    try:
        import threadpoolctl

        d["blas"] = [{k: p.get(k) for k in ("user_api", "internal_api", "version", "num_threads", "threading_layer", "architecture")}
                     for p in threadpoolctl.threadpool_info()]
    except Exception:
        d["blas"] = None
    return d


def cpu_baseline(cfg, coords, values, sample_pts, window=None, full=False):
    """The reference on the GPU box's host cores, next to the GPU number.

    OK2D with a named variogram: the reference's OWN compiled loop (lib/cok.pyx _c_exec_loop / _c_exec_loop_moving_window,
    built from /root/reference by oracle/build_ref.sh into oracle/_ref/ -- kind "reference"), i.e. backend='C', the >= 10x
    target.  Beside it (and alone for UK / 3-D, where the reference has no C backend): the NumPy / SciPy restatement of
    backend='vectorized' (oracle/kriging_oracle.py: same cdist / scipy.linalg.inv / np.dot calls -- kind "port"; the
    reference's Python package itself does not exist on the GPU box).  Protocol: a row slab of the config's own grid;
    OpenBLAS thread count swept over {1, 8, 16, 32, 64} on a small slab (dgemv on every core of a 256-thread host is
    oversubscribed) and the best kept; >= 3 repeats, best-of; fixed costs (matrix, inverse) timed on their own so that the
    steady-state rate and the full-grid projection can be separated.  Bounded run: `sample_pts` points (about 30 s);
    --cpu-protocol full: >= 16 384 points."""
    import scipy.linalg
    import threadpoolctl

    from oracle import kriging_oracle as ko
    from oracle import ref_c_loop as rc

    ndim = cfg["ndim"]
    st = ko.KrigingState(ndim=ndim, coords_orig=np.stack(coords, 1), values=values, model=cfg["model"],
                         params=internal_params(cfg["model"], cfg["params"]), scaling=[1.0] * (ndim - 1),
                         angle=[0.0] * (2 * ndim - 3), regional_linear=bool(cfg.get("rl")),
                         point_log=np.array(cfg["wells"]) if cfg.get("wells") else None)
    n_slab = max(16384, sample_pts) if full else sample_pts
    pts = row_slab(cfg, n_slab)
    n_slab = pts.shape[0]
    repeats = 3
    host = host_description()
    ncpu = os.cpu_count() or 1
    sweep = [t for t in (1, 8, 16, 32, 64) if t <= ncpu] or [1]
    use_c = rc.available() and ndim == 2 and not cfg.get("rl") and cfg["model"] != "hole-effect"
    npt_full = int(np.prod(cfg["grid"]))
    out = {"unit": "grid-points/s", "host": host, "slab_points": n_slab, "repeats": repeats,
           "protocol": "full (BASELINE.md section 3)" if full else "bounded (same protocol on a smaller slab)"}

    def best_of(fn, k):
        best, res = None, None
        for _ in range(k):
            t0 = time.perf_counter()
            res = fn()
            dt = time.perf_counter() - t0
            best = dt if best is None or dt < best else best
        return best, res

    if window:  # moving window: no N x N inverse, the fixed cost is the KD tree (inside the timed call)
        if use_c:
            sweep_res = {}
            small = pts[:min(n_slab, 2048)]
            for t in sweep:
                with threadpoolctl.threadpool_limits(limits=t):
                    dt, _ = best_of(lambda: rc.c_backend_moving_window(st, small, window), 1)
                sweep_res[t] = small.shape[0] / dt
            tbest = max(sweep_res, key=sweep_res.get)
            with threadpoolctl.threadpool_limits(limits=tbest):
                dt, (z, ss, _, t_loop) = best_of(lambda: rc.c_backend_moving_window(st, pts, window), repeats)
            out.update(value=n_slab / dt, cores=tbest, kind="reference", thread_sweep_points_per_s=sweep_res,
                       sample="%d-point row slab of the same grid, best of %d: cKDTree.query + PyKrige lib/cok.pyx "
                              "_c_exec_loop_moving_window (backend='C', n_closest_points=%d), %d BLAS threads" % (n_slab, repeats, window, tbest),
                       steady_state=n_slab / dt, full_grid_projection=n_slab / dt)
        else:
            sub = pts[:min(n_slab, 2048)]
            dt, (z, ss) = best_of(lambda: ko.solve_points_moving_window(st, sub, window), 1)
            pts = sub
            out.update(value=sub.shape[0] / dt, cores=1, kind="port",
                       sample="%d points, Python restatement of backend='loop' with n_closest_points=%d" % (sub.shape[0], window),
                       steady_state=sub.shape[0] / dt, full_grid_projection=sub.shape[0] / dt)
        return out, (pts, z, ss)

    # ---- dense path: fixed costs on their own -------------------------------------------------------------------
    t_mat, a = best_of(lambda: ko.kriging_matrix(st), 1)
    t_inv, a_inv = best_of(lambda: scipy.linalg.inv(a), 2)
    out["fixed_costs_s"] = {"matrix": t_mat, "inverse": t_inv}
    out["cond_1"] = float(np.linalg.norm(a, 1) * np.linalg.norm(a_inv, 1))
    # vectorized (port)
    t_vec, (zv, ssv) = best_of(lambda: ko.solve_points(st, pts, a_inv=a_inv), repeats)
    vec = {"value": n_slab / (t_mat + t_inv + t_vec), "steady_state": n_slab / t_vec, "kind": "port",
           "cores": max([p.get("num_threads", 1) for p in threadpoolctl.threadpool_info()] or [1]),
           "what": "NumPy/SciPy restatement of backend='vectorized' (oracle/kriging_oracle.py), 4096-point dgemm slabs, best of %d" % repeats,
           "full_grid_projection": npt_full / (t_mat + t_inv + npt_full * t_vec / n_slab)}
    out["vectorized"] = vec
    # backend='vectorized' of the reference ITSELF (ok.py:650-683, uk.py:922-1009, ok3d.py:624-657 as written upstream: the staged
    # package, oracle/ref_package.py) on the same slab through execute('grid', slab axes): kind "reference" for every config
    from oracle import ref_package as rp

    zv_ref = None
    if rp.available():
        try:
            pk = rp.import_reference(stub_statistics=True)
            rm = reference_model(pk, cfg, coords, values)
            slab_axes = row_slab_axes(cfg, n_slab)
            t_ref, (zr_, sr_) = best_of(lambda: rm.execute("grid", *slab_axes, backend="vectorized"), repeats)
            zv_ref, ssv_ref = np.ma.getdata(zr_).ravel(), np.ma.getdata(sr_).ravel()
            t_steady = max(t_ref - t_mat - t_inv, 1e-9)
            vec = {"value": n_slab / t_ref, "steady_state": n_slab / t_steady, "kind": "reference", "cores": vec["cores"],
                   "what": "the reference's own %s.execute('grid', row slab, backend='vectorized') (staged package, _find_statistics "
                           "stubbed), best of %d, incl. its matrix assembly and scipy.linalg.inv" % (type(rm).__name__, repeats),
                   "full_grid_projection": npt_full / (t_mat + t_inv + npt_full * t_steady / n_slab),
                   "port_vs_reference_max_abs_dz": float(np.abs(zv - zv_ref).max()),
                   "port_vs_reference_max_abs_dss": float(np.abs(ssv - ssv_ref).max()), "port": vec}
            out["vectorized"] = vec
            zv, ssv = zv_ref, ssv_ref
        except Exception as e:  # noqa: BLE001
            vec["reference_error"] = repr(e)[:200]
    if use_c:
        # thread sweep on the loop's hot operation itself -- one dgemv of the Fortran-ordered inverse per point
        # (cok.pyx:41, 71-83) -- so that the choice is not blurred by the inverse the native call repeats every time
        from scipy.linalg.blas import dgemv

        a_inv_f = np.asfortranarray(a_inv)
        bvec = np.ones(a_inv.shape[0])
        sweep_res = {}
        for t in sweep:
            with threadpoolctl.threadpool_limits(limits=t):
                dgemv(1.0, a_inv_f, bvec)
                reps = 24
                t0 = time.perf_counter()
                for _ in range(reps):
                    dgemv(1.0, a_inv_f, bvec)
                sweep_res[t] = reps / (time.perf_counter() - t0)
        tbest = max(sweep_res, key=sweep_res.get)
        del a_inv_f
        inv_t = {}
        with threadpoolctl.threadpool_limits(limits=tbest):
            inv_t[tbest], _ = best_of(lambda: scipy.linalg.inv(a), 2)
            best_total, best_loop, z, ss = None, None, None, None
            for _ in range(repeats):
                t0 = time.perf_counter()
                z, ss, t_loop = rc.c_backend(st, pts)
                dt = time.perf_counter() - t0
                if best_total is None or dt < best_total:
                    best_total, best_loop = dt, t_loop
        per_pt = max(best_loop - inv_t[tbest], 1e-9) / n_slab
        out.update(value=n_slab / (t_mat + best_loop), cores=tbest, kind="reference",
                   thread_sweep_dgemv_per_s=sweep_res,
                   sample="%d-point row slab of the same grid, best of %d: PyKrige lib/cok.pyx _c_exec_loop (backend='C') incl. "
                          "matrix assembly and its scipy.linalg.inv, %d OpenBLAS threads (best of the sweep)" % (n_slab, repeats, tbest),
                   steady_state=1.0 / per_pt,
                   full_grid_projection=npt_full / (t_mat + inv_t[tbest] + npt_full * per_pt),
                   c_vs_vectorized_max_abs_dz=float(np.abs(z - zv).max()),
                   c_vs_vectorized_max_abs_dss=float(np.abs(ss - ssv).max()))
    else:
        z, ss = zv, ssv
        out.update(value=vec["value"], cores=vec["cores"], kind=vec["kind"], sample="%d-point row slab of the same grid: %s" % (n_slab, vec["what"]),
                   steady_state=vec["steady_state"], full_grid_projection=vec["full_grid_projection"])
    return out, (pts, z, ss)


# ------------------------------------------------------------------------------------------------- live PMC traffic
def collect_traffic_live(argv_tail, kernel_prefix, timeout=240):
    """HBM-side bytes per launch of the dominant kernel: two rocprofv3 passes (FETCH_SIZE, WRITE_SIZE -- they do not fit
    one pass, MI355X_MICROARCH.md 'PMC slots') over ONE step of this very benchmark.  gfx950 correction (same guide,
    HBM section): FETCH_SIZE counts 128-byte requests as 64 bytes -> read bytes = 2 x FETCH_SIZE x 1024 (calibrated in
    round 1 on k_cvec / k_rhs, profiles/k_contract_traffic.json)."""
    exe = shutil.which("rocprofv3")
    if not exe:
        return None, "rocprofv3 not on PATH"
    res = {}
    tmp = tempfile.mkdtemp(prefix="mikpmc_", dir="/tmp")
    try:
        env = dict(os.environ, TMPDIR="/tmp", MIK_BENCH_INNER="1")
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, ctr)
            cmd = [exe, "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", d, "-o", "pmc", "--",
                   sys.executable, os.path.join(ROOT, "bench.py")] + argv_tail
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None, "rocprofv3 --pmc %s failed (rc %d): %s" % (ctr, r.returncode, (r.stderr or "")[-200:])
            tot, n = 0.0, 0
            for f in files:
                for row in csv.DictReader(open(f)):
                    if row["Kernel_Name"].startswith(kernel_prefix) and row["Counter_Name"] == ctr:
                        tot += float(row["Counter_Value"])
                        n += 1
            if n == 0:
                return None, "no %s dispatches in the %s pass" % (kernel_prefix, ctr)
            res[ctr] = (tot / n, n)
    except subprocess.TimeoutExpired:
        return None, "rocprofv3 pass timed out"
    except Exception as e:  # noqa: BLE001
        return None, "live PMC collection failed: %r" % (e,)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    fetch_kb, write_kb = res["FETCH_SIZE"][0], res["WRITE_SIZE"][0]
    return {"bytes_per_launch": 2.0 * fetch_kb * 1024.0 + write_kb * 1024.0, "FETCH_SIZE_KB_per_launch": fetch_kb,
            "WRITE_SIZE_KB_per_launch": write_kb, "launches_seen": res["FETCH_SIZE"][1]}, None


def grid_axes(cfg, n_gpus):
    """Axes of the weak-scaled grid: config's grid with its slowest axis n_gpus times as long (shard_points r = slab r of it)."""
    g = cfg["grid"]
    if cfg["ndim"] == 2:
        return [np.linspace(0.0, 1.0, g[0]), np.linspace(0.0, 1.0, g[1] * n_gpus)]
    return [np.linspace(0.0, 1.0, g[0]), np.linspace(0.0, 1.0, g[1]), np.linspace(0.0, 1.0, g[2] * n_gpus)]


def make_model(cfg, coords, values):
    """The drop-in class of the config (ok.py:187-206, uk.py:220-244, ok3d.py:198-219 constructor keywords)."""
    import pykrige_amd as pa

    kw = dict(variogram_model=cfg["model"], variogram_parameters=list(cfg["params"]))
    if cfg["ndim"] == 3:
        return pa.OrdinaryKriging3D(coords[0], coords[1], coords[2], values, **kw)
    if cfg.get("rl") or cfg.get("wells"):
        terms = (["regional_linear"] if cfg.get("rl") else []) + (["point_log"] if cfg.get("wells") else [])
        return pa.UniversalKriging(coords[0], coords[1], values, drift_terms=terms, point_drift=cfg.get("wells"), **kw)
    return pa.OrdinaryKriging(coords[0], coords[1], values, **kw)


def sparse_summary(t):
    """mik_timing's account of the range-aware contraction (last execute()): tiles contracted / tiles of the dense symmetric form."""
    return {"tiles_contracted": t["sparse_tiles"], "tiles_dense": t["sparse_tiles_dense"],
            "tile_fraction": t["sparse_tiles"] / max(1.0, t["sparse_tiles_dense"]),
            "offdiag_ktiles_contracted": t["sparse_ktiles"], "offdiag_ktiles_dense": t["sparse_ktiles_dense"],
            "ktile_fraction": t["sparse_ktiles"] / max(1.0, t["sparse_ktiles_dense"]),
            "list_kernels_ms": t["sparse_lists_ms"], "stations_in_hilbert_order": bool(t["stations_sorted"]),
            "tile_rows": ("eight gathered 16-row groups (k_contract_spg)" if t.get("sparse_rows") == 16
                          else "aligned blocks of 128 rows (k_contract_sp)"),
            "stations_per_list_tile": t.get("sparse_ktile", 16),  # 8 (round 5): a K step is a pair of list-adjacent 8-station tiles
            "triangle_products_16x16x128": t.get("sparse_diag_products", 0.0),
            "points_in_hilbert_order_per_launch": bool(t.get("points_sorted")), "sort_points_ms": t.get("sort_points_ms", 0.0)}


def sparse_kernel(t):
    return "k_contract_spg" if t.get("sparse_rows") == 16 else "k_contract_sp"


def golden_slab(cno):
    """tests/golden/fullsize/c<cno>.npz: a >= 16 384-point row slab of the config's own grid kriged by the REAL reference
    (oracle/make_golden_fullsize.py), with the stations it used (8 of them moved onto grid nodes)."""
    f = os.path.join(ROOT, "tests", "golden", "fullsize", "c%d.npz" % cno)
    with np.load(f, allow_pickle=False) as g:
        return {k: g[k] for k in g.files}


def other_config_line(cno, window=None, steps=3, warmup=1):
    """One of BASELINE's other configurations under the same clock as the headline (round-3 review: only config 2 was driver-timed):
    `steps` execute('grid') calls of the drop-in class after `warmup`, MIK_FACTOR_CACHE=0, no CPU leg, no PMC; then the stored
    reference slab of that configuration (tests/golden/fullsize) kriged by the same class and compared."""
    cfg = CONFIGS[cno]
    ndim = cfg["ndim"]
    coords, values = synth(cfg["seed"], cfg["n"], ndim)
    axes = grid_axes(cfg, 1)
    npt = int(np.prod([a.size for a in axes]))
    m = make_model(cfg, coords, values)
    hh = m._get_handle()
    kw = dict(backend="loop")
    if window:
        kw["n_closest_points"] = window
    for _ in range(warmup):
        m.execute("grid", *axes, **kw)
    hh.synchronize()
    acc = dict(contract_ms=0.0, contract_flops_executed=0.0, contract_launches=0, invert_ms=0.0, rhs_ms=0.0, predict_ms=0.0)
    t0 = time.perf_counter()
    for _ in range(steps):
        m.execute("grid", *axes, **kw)
        for k in acc:
            acc[k] += m.last_timing[k]
    hh.synchronize()
    dt = time.perf_counter() - t0
    last = dict(m.last_timing)
    M = cfg["n"] + (ndim if cfg.get("rl") else 0) + (len(cfg["wells"]) if cfg.get("wells") else 0) + 1
    line = {"workload": cfg["name"] + (", moving window n_closest_points=%d" % window if window else ""), "value": npt * steps / dt,
            "unit": "grid-points/s", "ms_per_step": dt / steps * 1e3, "steps": steps, "warmup": warmup, "grid_points": npt,
            "phases_ms_per_step": {"invert": acc["invert_ms"] / steps, "rhs": acc["rhs_ms"] / steps, "contract": acc["contract_ms"] / steps,
                                   "predict_total": acc["predict_ms"] / steps}}
    if window:
        nn = window + 1.0
        flops_pt = 2.0 / 3.0 * nn ** 3 + 2.0 * nn ** 2  # the reference's dgesv per point (cok.pyx:165)
        ach = flops_pt * npt * steps / (acc["contract_ms"] * 1e-3) / 1e12 if acc["contract_ms"] > 0 else 0.0
        line["roofline"] = {"bound": "mfma", "kernel": MW_KERNELS.get(last.get("mw_kernel"), "?"), "achieved": ach, "unit": "TFLOP/s",
                            "peak": FP64_MFMA_PEAK_TFLOPS, "frac": ach / FP64_MFMA_PEAK_TFLOPS, "algorithmic_flops_per_point": flops_pt}
    else:
        ach = acc["contract_flops_executed"] / (acc["contract_ms"] * 1e-3) / 1e12 if acc["contract_ms"] > 0 else 0.0
        useful = float(M) * M * npt * steps / (acc["contract_ms"] * 1e-3) / 1e12 if acc["contract_ms"] > 0 else 0.0
        line["roofline"] = {"bound": "mfma", "kernel": sparse_kernel(last) if last.get("sparse") else "k_contract", "achieved": ach,
                            "unit": "TFLOP/s", "peak": FP64_MFMA_PEAK_TFLOPS, "frac": ach / FP64_MFMA_PEAK_TFLOPS,
                            "useful_tflops": useful, "avg_launch_ms": acc["contract_ms"] / max(1, acc["contract_launches"])}
        if last.get("sparse"):
            line["roofline"]["sparse"] = sparse_summary(last)
    # parity on the stored reference slab
    try:
        import pykrige_amd as pa

        if window:
            with np.load(os.path.join(ROOT, "tests", "golden", "fullsize", "mw_c2.npz"), allow_pickle=False) as f:
                g = {k: f[k] for k in f.files}
            gm = pa.OrdinaryKriging(g["x"], g["y"], g["v"], variogram_model=str(g["model"]), variogram_parameters=g["params_user"].tolist())
            z, ss = gm.execute("grid", g["gridx"], g["gridy"], backend="loop", n_closest_points=window)
            zr, sr, src = g["z_k%d" % window], g["ss_k%d" % window], "tests/golden/fullsize/mw_c2.npz (reference backend='C')"
        else:
            g = golden_slab(cno)
            gcfg = dict(cfg)
            gm = make_model(gcfg, [g["x"], g["y"]] + ([g["zc"]] if ndim == 3 else []), g["v"])
            z, ss = gm.execute("grid", *([g["gridx"], g["gridy"]] + ([g["gridz"]] if ndim == 3 else [])), backend="loop")
            zr, sr, src = g["z"], g["ss"], "tests/golden/fullsize/c%d.npz (reference backend='vectorized')" % cno
        line["max_abs_dz"] = float(np.abs(np.ma.getdata(z) - zr).max())
        line["max_abs_dss"] = float(np.abs(np.ma.getdata(ss) - sr).max())
        line["parity_points"] = int(zr.size)
        line["parity_source"] = src
        gm._get_handle().close()
    except Exception as e:  # noqa: BLE001
        line["parity_error"] = repr(e)[:200]
    hh.close()
    return line


# ------------------------------------------------------------------------------------------------- parity over whole grids
# points per reference slab (whole rows / z planes: the reference's npt x N temporaries bd, b, x stay at a few GB) and, for
# config 5, the 4096 x 64 strip of the full 4096 x 4096 grid that straddles the cut between the slabs of GPUs 0 and 1 (row 512)
FULL_GRID = {2: dict(target=50000), 3: dict(target=40000), 4: dict(target=49152), 5: dict(target=32768, strip=(480, 544))}


def full_grid_axes(cno, strip=None):
    cfg = CONFIGS[cno]
    if cno == 5:
        r0, r1 = strip or FULL_GRID[5]["strip"]
        return [np.linspace(0.0, 1.0, 4096), np.linspace(0.0, 1.0, 4096)[r0:r1]]
    return grid_axes(cfg, 1)


def full_grid_parity(cno, budget_s=None, log=None, strip=None):
    """Every point of ONE execute('grid') of the drop-in class over the config's own grid (configs 2, 3, 4: the whole grid; config 5:
    a 4096 x 64 strip across the cut between two GPUs' slabs) against the REAL reference kriging the same grid slab by slab
    (oracle/full_grid.py; ok.py:650-683, uk.py:922-1009, ok3d.py:624-657 as written upstream).  `budget_s` bounds the CPU side: what
    fits is spread over the grid and `coverage` says how much it was."""
    from oracle import full_grid as fg
    from oracle import ref_package as rp

    cfg = CONFIGS[cno]
    coords, values = synth(cfg["seed"], cfg["n"], cfg["ndim"])
    axes = full_grid_axes(cno, strip)  # (strip: other rows of config 5's 4096 x 4096 grid, e.g. (0, 512) = the whole slab of GPU 0)
    m = make_model(cfg, coords, values)
    t0 = time.perf_counter()
    z, ss = m.execute("grid", *axes, backend="vectorized")
    t_gpu = time.perf_counter() - t0
    tm = dict(m.last_timing)
    m._get_handle().close()
    pk = rp.import_reference(stub_statistics=True)
    rm = reference_model(pk, cfg, coords, values)
    n = cfg["n"]
    extra = ((2 if cfg.get("rl") else 0) + (len(cfg["wells"]) if cfg.get("wells") else 0)) if cfg["ndim"] == 2 else 0
    res = {"workload": cfg["name"] + (" -- rows %d:%d of the 4096 x 4096 grid" % tuple(strip or FULL_GRID[5]["strip"]) if cno == 5 else ""),
           "grid": [int(a.size) for a in axes], "gpu_execute_s": t_gpu, "gpu_contraction": "range-aware" if tm.get("sparse") else "dense",
           "cond_1": fg.cond_1(rm, *((n, n + extra) if extra else (n,)))}
    # the reference side: several reference processes side by side (its _exec_vector is mostly single-threaded NumPy: one process leaves a 64-core
    # host idle), up to eight, as cores / 16 and memory allow (6 temporaries of npt x N doubles per slab, ok.py:669-681)
    need = 6 * 8 * FULL_GRID[cno]["target"] * (n + 1) * 1.5
    try:
        avail = [int(ln.split()[1]) * 1024 for ln in open("/proc/meminfo") if ln.startswith("MemAvailable")][0]
    except Exception:  # noqa: BLE001
        avail = 16 << 30
    workers = int(max(1, min(8, (os.cpu_count() or 1) // 16, avail * 0.5 // need)))
    if os.environ.get("MIK_FULLGRID_WORKERS"):
        workers = max(1, int(os.environ["MIK_FULLGRID_WORKERS"]))
    res.update(fg.compare(rm, z, ss, axes, FULL_GRID[cno]["target"], budget_s=budget_s, log=log, recipe=(cfg, coords, values), workers=workers))
    if cno == 2 and rp.c_available():
        # the reference's two CPU paths against each other on two rows of the same grid (cok.pyx:56-94 vs ok.py:650-683): the
        # size of the disagreement the reference itself lives with, next to the GPU-vs-reference one
        mid = axes[1].size // 2
        sl = [axes[0], axes[1][mid:mid + 2]]
        zc, sc = rm.execute("grid", *sl, backend="C")
        zv, sv = rm.execute("grid", *sl, backend="vectorized")
        res["reference_c_vs_vectorized_max_abs_dz"] = float(np.abs(np.ma.getdata(zc) - np.ma.getdata(zv)).max())
        res["reference_c_vs_vectorized_max_abs_dss"] = float(np.abs(np.ma.getdata(sc) - np.ma.getdata(sv)).max())
    res["host_cpus"] = os.cpu_count()
    res["tolerance"] = {"z": 1e-8, "ss": 1e-6}
    res["ok"] = bool(res["max_abs_dz"] <= 1e-8 and res["max_abs_dss"] <= 1e-6)
    return res


def compact_other(line):
    """The scalars of an other_configs line that scalars_for_driver() puts under `config` as c3_*, c4_*, c5_*, mw_k10_*, mw_k100_*."""
    roof = line.get("roofline") or {}
    c = {"value": line.get("value"), "ms_per_step": line.get("ms_per_step"), "frac": roof.get("frac"), "kernel": roof.get("kernel"),
         "max_abs_dz": line.get("max_abs_dz"), "max_abs_dss": line.get("max_abs_dss")}
    inv = (line.get("phases_ms_per_step") or {}).get("invert")
    if inv is not None:
        c["invert_ms"] = inv
    for k in ("error", "parity_error"):
        if line.get(k):
            c[k] = line[k]
    return c


def compact_multi_gpu(mg):
    """The multi-GPU facts scalars_for_driver() puts under `config`: which exchange ran, on how many RCCL ranks, and what every device's prediction took."""
    per = mg.get("per_device_predict_ms")
    if per is None and mg.get("ranks"):
        per = [r.get("predict_ms") for r in mg["ranks"]]
    return {"rccl_ranks": mg.get("rccl_ranks"), "exchange_path": mg.get("exchange_path"), "per_device_predict_ms": per,
            "exchange_ms": mg.get("exchange_ms"), "exchange_wait_ms": mg.get("exchange_wait_ms"), "exchange_bytes": mg.get("exchange_bytes"),
            "exchange_fallbacks": mg.get("exchange_fallbacks"), "exchange_note": mg.get("exchange_note")}


SHORT = {"config3": "c3", "config4": "c4", "config5": "c5", "moving_window_k10": "mw_k10", "moving_window_k100": "mw_k100"}


def scalars_for_driver(out):
    """The driver's BENCH record keeps the SCALAR keys of `config` and nothing else of the line (round-5 review: the nested
    other_configs / multi_gpu / phases dicts it was given were dropped).  Everything the record must hold goes here as flat scalar
    keys; the key set is pinned by tests/test_bench_host.py.  The full objects stay at the top level of the line."""
    c = {}
    for k, v in (out.get("phases_ms_per_step") or {}).items():
        c["phase_%s_ms" % k] = v
    for key, line in (out.get("other_configs") or {}).items():
        p = SHORT.get(key, key)
        for f, v in compact_other(line).items():
            if v is not None or f in ("value", "max_abs_dz", "max_abs_dss"):
                c["%s_%s" % (p, f)] = v
    mg = out.get("multi_gpu")
    if mg:
        cm = compact_multi_gpu(mg)
        for k in ("rccl_ranks", "exchange_path", "exchange_ms", "exchange_wait_ms", "exchange_bytes", "exchange_fallbacks", "exchange_note"):
            c[k] = cm.get(k)
        per = [p for p in (cm.get("per_device_predict_ms") or []) if p is not None]
        if per:
            c["predict_ms_slowest_device"], c["predict_ms_fastest_device"] = max(per), min(per)
            c["per_device_predict_ms"] = ",".join("%.2f" % p for p in per)
    for k, v in (out.get("factor_exchange_trial") or {}).items():
        c["trial_" + k] = v if isinstance(v, (int, float, str, bool)) or v is None else json.dumps(v)
    cb = out.get("cpu_baseline")
    if cb:
        for k in ("value", "kind", "cores", "cond_1", "gpu_vs_cpu_max_abs_dz", "gpu_vs_cpu_max_abs_dss", "slab_points"):
            c["cpu_" + k if not k.startswith("gpu_") else k] = cb.get(k)
    fgp = out.get("full_grid_parity")
    if fgp:
        for k in ("points_checked", "points_total", "coverage", "max_abs_dz", "max_abs_dss", "cond_1", "reference_points_per_s", "ok", "error"):
            if k in fgp:
                c["c2_fullgrid_" + k] = fgp[k]
    roof = out.get("roofline") or {}
    for k in ("kernel", "avg_launch_ms", "traffic", "algorithmic_bytes_per_launch"):
        if roof.get(k) is not None:
            c["roofline_" + k] = roof[k]
    return c


MW_KERNELS = {1: "k_mw_chol", 2: "k_mw_solve", 3: "k_mw_solve_big", 4: "k_mw_chol_blocked"}
FACTOR_PATHS = {1: "spd-shift block sweep", 2: "pivoted block gauss-jordan", 3: "caller-supplied inverse", 4: "device pseudo-inverse (jacobi)",
                5: "deflated inverse (pseudo-inverse of duplicated stations)", 6: "deflated inverse (numerically found null space)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)  # (the part's clock settles over the first second of load)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-sample", type=int, default=4096, help="points of the bounded CPU slab")
    ap.add_argument("--cpu-protocol", choices=["bounded", "full"], default="full",
                    help="full (default) = BASELINE.md section 3 to the letter (>= 16 384-point slab, best of 3, thread sweep): about a "
                         "minute of CPU time; bounded = the same protocol on --cpu-sample points")
    ap.add_argument("--pmc", choices=["auto", "on", "off"], default="auto",
                    help="collect roofline.traffic live with two rocprofv3 --pmc passes of one step (auto: when N = 1)")
    ap.add_argument("--exchange", choices=sorted(EXCHANGE_CODES), default=os.environ.get("MIK_BENCH_EXCHANGE", "auto"),
                    help="how the inverted matrix reaches the other GPUs (auto: RCCL broadcast, else peer copies)")
    ap.add_argument("--symmetric", type=int, default=None)
    ap.add_argument("--chunk", type=int, default=None)
    ap.add_argument("--factor", choices=["auto", "sweep", "lu"], default=None, help="force the inverse path")
    ap.add_argument("--moving-window", type=int, default=None, metavar="K",
                    help="time moving-window kriging (n_closest_points=K) on the same workload instead (not the headline metric)")
    ap.add_argument("--sparse", type=int, default=None, choices=[-1, 0, 1, 2],
                    help="library option 'sparse' (range-aware contraction of the spherical model; default: the library's, on)")
    ap.add_argument("--sparse-rows", type=int, default=None, choices=[-1, 16, 128],
                    help="library option 'sparse_rows' (range-aware contraction: tiles of eight gathered 16-row groups, or aligned "
                         "128-row blocks; default: the library's, 16)")
    ap.add_argument("--sparse-lanes", type=int, default=None, choices=[1, 2], help="library option 'sparse_lanes' (default: the library's, 2)")
    ap.add_argument("--sort-points", type=int, default=None, choices=[-1, 0, 1],
                    help="library option 'sort_points' (range-aware contraction over the points of every launch in Hilbert-curve order; "
                         "default: the library's, on)")
    ap.add_argument("--no-trials", action="store_true", default=os.environ.get("MIK_BENCH_TRIALS", "1") == "0",
                    help="device groups: skip the exchange / overlap trials before the timed loop (also MIK_BENCH_TRIALS=0)")
    ap.add_argument("--pretrial-budget", type=float, default=float(os.environ.get("MIK_BENCH_PRETRIAL_BUDGET", "60")), metavar="S",
                    help="device groups: wall-clock seconds the trials before the timed loop may take in total (default 60); what "
                         "does not fit is skipped and config.factor_exchange_trial says so")
    ap.add_argument("--full-parity", nargs="?", const="2,4,3,5", default=None, metavar="CONFIGS",
                    help="instead of timing: krige the WHOLE grid of each listed config (default 2,4,3,5; config 5: a 4096 x 64 strip across "
                         "the cut between two GPUs' slabs) with the drop-in class and with the staged reference (backend='vectorized', "
                         "slab by slab on the host cores) and compare every point at 1e-8 / 1e-6; one JSON line per config")
    ap.add_argument("--full-parity-budget", type=float, default=None, metavar="S",
                    help="--full-parity: wall-clock bound of the reference side per config (default: none, the whole grid)")
    ap.add_argument("--full-grid-budget", type=float, default=float(os.environ.get("MIK_BENCH_FULLGRID_BUDGET", "90")), metavar="S",
                    help="default run: seconds the reference may spend on the headline's WHOLE-grid parity check after the timed steps "
                         "(slabs spread over the grid; config.c2_fullgrid_* say how many points that covered; 0 = skip)")
    ap.add_argument("--no-other", action="store_true",
                    help="default run (config 2, 1 GPU): do not time BASELINE configs 3-5 and the moving window after the headline")
    args = ap.parse_args()
    inner = os.environ.get("MIK_BENCH_INNER") == "1"  # a rocprofv3 pass of collect_traffic_live: no CPU leg, no recursion
    if args.full_parity:
        os.environ["MIK_FACTOR_CACHE"] = "0"
        bad = 0
        for key in args.full_parity.split(","):
            # "5w" = config 5 with the WHOLE slab of GPU 0 (rows 0:512 of the 4096 x 4096 grid, 2.1 M points: about 8 minutes of reference time)
            cno, strip = (5, (0, 512)) if key == "5w" else (int(key), None)
            res = full_grid_parity(cno, budget_s=args.full_parity_budget, log=lambda m: print(m, file=sys.stderr, flush=True), strip=strip)
            bad += not res["ok"]
            print(json.dumps(dict(full_grid_parity="config" + key, **res)), flush=True)
        sys.exit(1 if bad else 0)

    # Keep stdout clean for the ONE JSON line: RCCL prints banners from C++ to fd 1, so fd 1 points at
    # stderr while the benchmark runs and is restored just before the JSON is printed.
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)

    def emit(line):
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        print(line, flush=True)
        os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))  # > 1: launched by torch.distributed.run, one process per GPU
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    group = 1  # devices spanned by this process's handle
    if world > 1:
        args.gpus = world
    elif args.gpus > 1:
        group = args.gpus  # plain `python bench.py --gpus N`: single process, the library's device group
    n_gpus = world * group

    # A line ALWAYS comes out.  Every wait of the multi-GPU paths is bounded inside the library (include/mikrige.h, "Bounded
    # waits"); this is the last line of defence for anything else that stalls: after MIK_BENCH_DEADLINE seconds rank 0 prints
    # an error line (value null) and the process exits.
    import threading

    progress = {"stage": "start", "done": False}

    def deadline_watch(limit):
        t_end = time.time() + limit
        while time.time() < t_end:
            if progress["done"]:
                return
            time.sleep(0.5)
        if rank == 0:
            emit(json.dumps({"metric": "kriged grid-points/sec (z + sigma^2)", "value": None, "unit": "grid-points/s", "n_gpus": n_gpus,
                             "steps": args.steps, "warmup": args.warmup, "higher_is_better": True, "scaling": "weak",
                             "error": "bench.py did not finish within %.0f s; last stage: %s" % (limit, progress["stage"])}))
        os._exit(3)

    if not inner:
        threading.Thread(target=deadline_watch, args=(float(os.environ.get("MIK_BENCH_DEADLINE", "1500")),), daemon=True).start()

    pg = None
    if world > 1:
        # Host-side rendezvous / barrier / max-over-ranks only.  The launcher's env (RANK, WORLD_SIZE, MASTER_*) is used
        # through a small TCP group so that the process holds ONE HIP runtime and ONE RCCL (the ROCm install's, the ones
        # libmikrige.so links and dlopens); torch's gloo group is the fallback if that rendezvous cannot be set up.
        from pykrige_amd.dist import SocketGroup, _TorchGroup

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        try:
            if os.environ.get("MIK_BENCH_HOSTPG", "socket") != "socket":
                raise RuntimeError("torch group requested")
            pg = SocketGroup(rank=rank, world=world)
        except Exception as e:  # noqa: BLE001
            print("bench: socket group unavailable (%r); using torch.distributed gloo" % (e,), file=sys.stderr)
            import torch.distributed as tdist

            tdist.init_process_group(backend="gloo", rank=rank, world_size=world)
            pg = _TorchGroup()

    from pykrige_amd import _lib  # raises if libmikrige.so is missing: no CPU fallback

    cfg = CONFIGS[args.config]
    ndim = cfg["ndim"]
    coords, values = synth(cfg["seed"], cfg["n"], ndim)
    axes = grid_axes(cfg, n_gpus)  # the whole weak-scaled grid; the library (device group) / the executor (ranks) cuts the slabs
    npt_total = int(np.prod([a.size for a in axes]))
    npt_rank = npt_total // world

    ndev = _lib.load().mik_device_count()
    aliased = group > max(ndev, 1)
    os.environ["MIK_FACTOR_CACHE"] = "0"  # every execute() assembles and inverts the matrix, as the reference does (ok.py:898, 663)
    os.environ["MIK_DEVICE"] = str(local_rank % max(ndev, 1))  # one GPU per rank on a real node; wraps only on a 1-GPU test box
    if aliased:
        os.environ["MIK_ALIAS_DEVICES"] = "1"
    if group > 1:
        _lib.set_devices(group)
    kw = args.moving_window

    def make():
        m = make_model(cfg, coords, values)
        hh = m._get_handle()
        if group > 1:
            hh.set_option("exchange", EXCHANGE_CODES[args.exchange])
        if args.symmetric is not None:
            hh.set_option("symmetric", args.symmetric)
        if args.chunk is not None:
            hh.set_option("chunk", args.chunk)
        if args.factor is not None:
            hh.set_option("factor", {"auto": 0, "sweep": 1, "lu": 2}[args.factor])
        if args.sparse is not None:
            hh.set_option("sparse", args.sparse)
        if args.sparse_rows is not None:
            hh.set_option("sparse_rows", args.sparse_rows)
        if args.sort_points is not None:
            hh.set_option("sort_points", args.sort_points)
        if args.sparse_lanes is not None:
            hh.set_option("sparse_lanes", args.sparse_lanes)
        return m, hh

    progress["stage"] = "create the kriging object and its device handle"
    model, h = make()
    exe_kw = dict(backend="loop")
    if kw:
        exe_kw["n_closest_points"] = kw

    # How the inverted matrix reaches every GPU.  The north_star's design is the default and the one timed: the leader
    # factors, ONE broadcast over RCCL/xGMI.  The alternatives are measured beside it, outside the timed region, and
    # printed (factor_exchange_trial): peer copies shaped as scatter + all-gather, and no exchange at all (every GPU
    # factors the identical matrix; the others wait for the leader's factorisation anyway, so this is a wall-clock tie at
    # best for the broadcast and costs N-1 redundant O(M^3) factorisations of energy).  Every trial is bounded by the
    # library's own limits (a stalled RCCL call is abandoned, reported, and the next path runs).
    exchange, trial, executor = "none", None, None
    if group > 1 and not kw and args.no_trials:
        trial = {"skipped": "--no-trials / MIK_BENCH_TRIALS=0: the default exchange path is timed as it comes"}
    elif group > 1 and not kw:
        # ONE wall-clock budget over everything that runs before the timed loop (the first real multi-GPU node meets code that has
        # only ever run on aliased devices: worst case every trial waits out a limit).  When it is spent the remaining trials are
        # skipped, the default path is timed, and the line says so.
        progress["stage"] = "exchange trials of the device group"
        t_trials = time.perf_counter()
        budget = max(0.0, args.pretrial_budget)
        trial = {"budget_s": budget}

        def budget_left():
            return budget - (time.perf_counter() - t_trials)

        model._set_problem(h)
        for name in ("rccl", "peer", "redundant"):
            if budget_left() <= 0.0:
                trial.setdefault("skipped_for_budget", []).append(name)
                continue
            try:
                h.set_option("exchange", EXCHANGE_CODES[name])
                h.set_option("async_exchange", 0)  # the trial times factor + exchange as one blocking call
                best = None
                for _ in range(2):  # the first round pays RCCL's communicator / stream set-up and the buffer allocations
                    t0 = time.perf_counter()
                    h.factor()
                    dt = time.perf_counter() - t0
                    best = dt if best is None else min(best, dt)
                trial[name + "_factor_plus_exchange_ms"] = best * 1e3
                trial[name + "_exchange_ms"] = h.timing()["exchange_ms"]
            except Exception as e:  # noqa: BLE001
                trial[name + "_error"] = repr(e)[:300]
        h.set_option("exchange", EXCHANGE_CODES[args.exchange])
        h.set_option("async_exchange", 1)
        # Should the leader krige its slab WHILE an RCCL broadcast is in flight?  The broadcast's root needs compute units that the
        # leader's persistent contraction launch holds, so the library's default (1) joins an RCCL exchange first and overlaps only
        # copy-engine transfers; 2 overlaps any.  Measured here on the hardware at hand, outside the timed region; the faster one runs.
        progress["stage"] = "overlap trial of the device group"
        try:
            if budget_left() <= 0.0:
                trial.setdefault("skipped_for_budget", []).append("overlap")
                raise StopIteration
            model.execute("grid", *axes, **exe_kw)
            if model.last_timing["exchange_path"] == 1 and budget_left() <= 0.0:
                trial.setdefault("skipped_for_budget", []).append("overlap")
            elif model.last_timing["exchange_path"] == 1:
                ov = {}
                for mode in (1, 2):
                    h.set_option("async_exchange", mode)
                    model.execute("grid", *axes, **exe_kw)
                    t0 = time.perf_counter()
                    model.execute("grid", *axes, **exe_kw)
                    ov[mode] = (time.perf_counter() - t0) * 1e3
                pick = min(ov, key=ov.get)
                h.set_option("async_exchange", pick)
                trial["execute_ms_leader_waits_for_rccl"], trial["execute_ms_leader_overlaps_rccl"] = ov[1], ov[2]
                trial["async_exchange_used"] = pick
        except StopIteration:
            pass
        except Exception as e:  # noqa: BLE001
            trial["overlap_trial_error"] = repr(e)[:300]
        trial["spent_s"] = time.perf_counter() - t_trials
    elif world > 1:
        progress["stage"] = "process group / RCCL set-up of the ranks"
        from pykrige_amd.dist import ShardedExecutor

        # rank r kriges slab r and keeps it (gather='local': "each GPU writes its slab"); rank 0 factors, the library
        # broadcasts over RCCL (bounded: MIK_RCCL_INIT_TIMEOUT / MIK_RCCL_BCAST_TIMEOUT), checksums are compared, and any
        # failure degrades to every rank factoring for itself
        executor = ShardedExecutor(model, group=pg, use_rccl=(args.exchange != "redundant") and not kw, gather="local")

    def sync():
        h.synchronize()  # every device of the handle and its result copies idle (the calls already block; explicit bracket)
        if pg is not None:
            pg.barrier()

    tsum = dict(assemble_ms=0.0, invert_ms=0.0, rhs_ms=0.0, contract_ms=0.0, predict_ms=0.0, contract_launches=0,
                contract_flops_executed=0.0, exchange_ms=0.0, exchange_wait_ms=0.0)
    last = {}

    def step(record):
        # ONE execute() of the drop-in class, host arrays in -> host arrays out (SURVEY 8d "wall-clock of execute() end-to-end"):
        # front matter, mik_set_problem, K1 + K2 [+ exchange], the grid generated on the device from its axes, K3 for every
        # point, z and sigma^2 into page-locked memory chunk by chunk, copy into the returned arrays, back matter
        if executor is not None:
            out = executor.execute("grid", *axes, **exe_kw)
            t = h.timing() if record else None
        else:
            out = model.execute("grid", *axes, **exe_kw)
            t = model.last_timing
        if record:
            for k in ("assemble_ms", "invert_ms", "exchange_ms", "exchange_wait_ms", "rhs_ms", "contract_ms", "predict_ms", "contract_launches",
                      "contract_flops_executed"):
                tsum[k] += t[k]
            last.update(t)
        return out

    progress["stage"] = "warm-up steps"
    for _ in range(args.warmup):
        step(False)
    sync()
    progress["stage"] = "timed steps"
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step(True)
    sync()
    dt = time.perf_counter() - t0
    if pg is not None:
        dt = pg.all_reduce_max(dt)
    if executor is not None:
        exchange = executor.exchange
    elif group > 1 and not kw:
        exchange = EXCHANGE_NAMES[last["exchange_path"]]
        note = h.exchange_note()
        if note:
            exchange += " (" + note + ")"

    if rank == 0:
        progress["stage"] = "report"
        K = args.steps
        M = cfg["n"] + (ndim if cfg.get("rl") else 0) + (len(cfg["wells"]) if cfg.get("wells") else 0) + 1
        value = npt_total * K / dt
        launches = max(1, int(tsum["contract_launches"]))  # of the leader device (its slab = total / n_gpus points)
        avg_launch_s = tsum["contract_ms"] * 1e-3 / launches
        pts_per_launch = (npt_total / n_gpus) * K / launches
        if kw:
            # dominant kernel: the per-point (k+1) x (k+1) solves.  Algorithmic work per point = what the reference's dgesv
            # does (cok.pyx:165): 2/3 n^3 + 2 n^2 flops, n = k + 1; bytes: k x 12 (neighbour index + distance) in, 16 out.
            nn = kw + 1.0
            algo_flops_pt = 2.0 / 3.0 * nn ** 3 + 2.0 * nn ** 2
            achieved = algo_flops_pt * pts_per_launch / avg_launch_s / 1e12 if avg_launch_s > 0 else 0.0
            kname = MW_KERNELS.get(last.get("mw_kernel"), "k_mw_chol")  # the solver that ran (mik_timing.mw_kernel)
            roof = {"bound": "mfma", "kernel": kname, "achieved": achieved, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": achieved / FP64_MFMA_PEAK_TFLOPS, "traffic": None,
                    "note": "fp64 compute roof (vector = matrix rate on this part): %.0f flop per point against %d bytes; the kernel "
                            "is a register-tiled elimination with one LDS exchange per step -- issue / latency bound, not MFMA work"
                            % (algo_flops_pt, 12 * kw + 16),
                    "avg_launch_ms": avg_launch_s * 1e3, "algorithmic_flops_per_point": algo_flops_pt,
                    "algorithmic_bytes_per_point": 12 * kw + 16}
            metric = "kriged grid-points/sec (z + sigma^2), moving window n_closest_points=%d, %s" % (kw, cfg["name"])
            config = {"workload": cfg["name"], "n_closest_points": kw, "stations": cfg["n"], "grid_points_total": npt_total}
            kernel_prefix = ("void mik::" + kname + "<") if kname in ("k_mw_chol", "k_mw_solve") else "mik::" + kname
        else:
            algo_flops_per_launch = 2.0 * M * M * pts_per_launch  # 2 M^2 per point for w = A_inv . b
            effective = algo_flops_per_launch / avg_launch_s / 1e12 if avg_launch_s > 0 else 0.0
            executed = tsum["contract_flops_executed"] / (tsum["contract_ms"] * 1e-3) / 1e12 if tsum["contract_ms"] > 0 else 0.0
            roof = {"bound": "mfma", "kernel": sparse_kernel(last) if last.get("sparse") else "k_contract",
                    "achieved": executed, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": executed / FP64_MFMA_PEAK_TFLOPS,
                    "peak_measured": FP64_MFMA_MEASURED_TFLOPS, "frac_of_measured_peak": executed / FP64_MFMA_MEASURED_TFLOPS,
                    "effective_tflops": effective, "effective_frac": effective / FP64_MFMA_PEAK_TFLOPS,
                    "useful_tflops": effective / 2.0, "useful_frac": effective / 2.0 / FP64_MFMA_PEAK_TFLOPS,
                    "traffic": None, "traffic_unit": "bytes per launch (HBM-side: 2 x FETCH_SIZE + WRITE_SIZE, gfx950 correction)",
                    "note": "achieved / frac = flops the kernel EXECUTES (symmetric half product, ~M^2 per point) over the fp64 matrix "
                            "peak: a true fraction.  effective_* = the reference's 2 M^2 flops per point (SURVEY 8d) over the same "
                            "time: the rate a full product would need to match this kernel; it can exceed the peak.  useful_* = M^2 "
                            "flops per point (the quadratic form over one triangle) over the same time: <= achieved, and the "
                            "number to compare across kernel versions (round 3's triangular diagonal blocks issue 2 % fewer "
                            "flops for the same result: achieved went down, useful_* and points/s went up).",
                    "avg_launch_ms": avg_launch_s * 1e3, "launches_per_step": launches / K,
                    "algorithmic_flops_per_point": 2.0 * M * M, "useful_flops_per_point": float(M) * M,
                    "executed_flops_per_point": tsum["contract_flops_executed"] / max(1.0, pts_per_launch * launches),
                    "flops_note": "algorithmic = the reference's w = A_inv.b (2 M^2, ok.py:679); useful = the quadratic form b^T A_inv b "
                                  "over one triangle (M^2); executed = what the kernel issues: useful + the mirrored halves of the 16 x 16 diagonal "
                                  "squares (option tri = 1, the default; of the whole 128 x 128 diagonal blocks with tri = 0)"}
            if last.get("sparse"):
                roof["sparse"] = sparse_summary(last)
                roof["note"] += ("  RANGE-AWARE CONTRACTION (spherical model): sigma^2 = 2 s - delta^T A_inv delta over the tiles that hold a "
                                 "nonzero of delta = b + s u only; achieved / frac still count the flops EXECUTED over the kernel's time -- "
                                 "shorter K loops, so frac is below the dense kernel's while points/s is a multiple of it; useful_* keeps "
                                 "the dense M^2 count per point (the cross-version number: it may exceed the peak here).")
            metric = ("kriged grid-points/sec (z + sigma^2), OK2D N=5000 on 1000x1000 grid" if args.config == 2
                      else "kriged grid-points/sec (z + sigma^2), " + cfg["name"])
            config = {"workload": cfg["name"], "stations": cfg["n"], "matrix_order": M,
                      "grid_points_per_gpu": npt_total // n_gpus, "grid_points_total": npt_total, "variogram": cfg["model"],
                      "variogram_parameters": ",".join(str(p) for p in cfg["params"]), "factor_exchange": exchange,
                      "factor_path": FACTOR_PATHS.get(last.get("factor_path"), "?"),
                      "symmetric_contraction": bool(last.get("symmetric"))}
            kernel_prefix = "void mik::k_contract"
        config["timed_call"] = ("%s.execute('grid', axes, backend='loop'%s): host arrays in, host arrays out; the grid is generated on the "
                                "device from its axes (mik_set_grid), every call assembles and inverts the matrix (MIK_FACTOR_CACHE=0, as "
                                "the reference does)" % (type(model).__name__, ", n_closest_points=%d" % kw if kw else ""))
        config["launch"] = ("one process per GPU (torch.distributed.run), pykrige_amd.dist.ShardedExecutor, gather='local'" if world > 1 else
                            "one process, device group of %d%s" % (group, " ALIASED onto %d physical GPU(s)" % ndev if aliased else "")
                            if group > 1 else "one process, one GPU")
        out = {"metric": metric, "value": value, "unit": "grid-points/s", "n_gpus": n_gpus, "steps": K, "warmup": args.warmup,
               "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f64", "data": "synthetic", "config": config, "roofline": roof,
               "phases_ms_per_step": {"assemble": tsum["assemble_ms"] / K, "invert": tsum["invert_ms"] / K,
                                      "exchange": tsum["exchange_ms"] / K, "exchange_not_overlapped": tsum["exchange_wait_ms"] / K,
                                      "rhs": tsum["rhs_ms"] / K, "contract": tsum["contract_ms"] / K, "predict_total": tsum["predict_ms"] / K}}
        if trial:
            out["factor_exchange_trial"] = trial
        if last.get("sparse"):
            config["phases_note"] = ("range-aware contraction on two launch lanes (option sparse_lanes 2): the right-hand-side / list kernels of one "
                                     "launch run beside the other lane's contraction; rhs and contract are per-kernel HIP-event sums and may "
                                     "add up to more than predict_total (the wall time of the prediction on the device)")
        dev_ms = (tsum["assemble_ms"] + tsum["invert_ms"] + tsum["exchange_wait_ms"] + tsum["predict_ms"]) / K
        out["host_overhead"] = {"vs_device_phases_ms": dt / K * 1e3 - dev_ms,
                                "what": "ms_per_step / frac (filled in below): execute() minus the `resident` step (mik_factor + mik_predict on "
                                        "resident points), both wall-clock -- what the Python front / back matter, mik_set_problem, mik_set_grid "
                                        "and the hand-over of the results cost.  vs_device_phases_ms: execute() minus the device phases by HIP "
                                        "events (assemble + invert + awaited exchange + the slowest device's predict); not meaningful when "
                                        "several group members share one GPU"}
        # ---- the multi-GPU path describes itself (every --gpus N)
        if group > 1:
            per_dev = [h.device_timing(i) for i in range(group)]
            out["multi_gpu"] = {"devices": [t["reserved"] for t in per_dev], "per_device_predict_ms": [t["predict_ms"] for t in per_dev],
                                "exchange_path": EXCHANGE_NAMES.get(last.get("exchange_path"), "?"), "rccl_ranks": last.get("rccl_ranks"),
                                "exchange_fallbacks": last.get("exchange_fallbacks"), "exchange_note": h.exchange_note(),
                                "exchange_ms": last.get("exchange_ms"), "exchange_wait_ms": last.get("exchange_wait_ms"),
                                "exchange_bytes": last.get("exchange_bytes"),  # per member: the packed upper block triangle of the inverse + c
                                "timeouts_s": {"rccl_init": float(os.environ.get("MIK_RCCL_INIT_TIMEOUT", "120")),
                                               "rccl_bcast": float(os.environ.get("MIK_RCCL_BCAST_TIMEOUT", "30")),
                                               "peer": float(os.environ.get("MIK_PEER_TIMEOUT", "30"))}}
            out["per_device_predict_ms"] = out["multi_gpu"]["per_device_predict_ms"]
        elif world > 1:
            allp = pg.all_gather_object({"rank": rank, "device": int(os.environ["MIK_DEVICE"]), "predict_ms": last.get("predict_ms")})
            out["multi_gpu"] = {"ranks": allp, "exchange_path": exchange, "rccl_ranks": world if exchange == "rccl_bcast" else 0,
                                "exchange_bytes": last.get("exchange_bytes"), "exchange_ms": getattr(executor, "exchange_ms", None)}
        # ---- roofline.traffic: live PMC passes over one step of this benchmark, else the committed profile (labelled)
        if n_gpus == 1 and not inner and args.pmc != "off":
            progress["stage"] = "live PMC passes (rocprofv3)"
            tail = ["--steps", "1", "--warmup", "0", "--no-cpu", "--pmc", "off", "--config", str(args.config)]
            if kw:
                tail += ["--moving-window", str(kw)]
            for flag, val in (("--symmetric", args.symmetric), ("--chunk", args.chunk), ("--factor", args.factor)):
                if val is not None:
                    tail += [flag, str(val)]
            live, why = collect_traffic_live(tail, kernel_prefix)
            if live:
                roof["traffic"] = live["bytes_per_launch"]
                roof["traffic_source"] = "live: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE, two passes of one step of this run"
                roof["traffic_counters"] = live
            else:
                roof["traffic_source"] = "live collection failed: " + str(why)
        if roof.get("traffic") is None and not kw:
            try:
                tj = json.load(open(os.path.join(ROOT, "profiles", "k_contract_traffic.json")))
                if tj["workload"] == cfg["name"] and last.get("symmetric"):
                    roof["traffic"] = tj["hbm_bytes_per_launch"] * (pts_per_launch / tj["points_per_launch"])
                    roof["traffic_source"] = ("from_profile (NOT collected in this run): " + tj["source"]
                                              + ("; " + roof["traffic_source"] if roof.get("traffic_source") else ""))
            except Exception:
                pass
        if not kw and not last.get("sparse") and last.get("symmetric"):
            # how much of `traffic` is HBM: no Infinity-Cache hit / miss counter exists on gfx950, so the split is made by L2-miss latency
            # in a run of its own (scripts/gpu_hbm_split.sh) and quoted from its tracked summary, scaled to this launch's points
            try:
                hj = json.load(open(os.path.join(ROOT, "profiles", "k_contract_hbm_split.json")))
                if hj["workload"] == cfg["name"]:
                    sc = pts_per_launch / hj["points_per_launch"]
                    roof["traffic_hbm"] = [v * sc for v in hj["traffic_hbm_bytes_per_launch_by_latency"]]
                    roof["traffic_hbm_source"] = ("from_profile (NOT collected in this run; lower .. upper estimate, bytes per launch): " + hj["source"]
                                                  + "; the rest of `traffic` are Infinity-Cache hits")
            except Exception:
                pass
        if not kw:
            # compulsory bytes of one launch: the inverse once, the RHS panel (8 M per point) in, 8 M/128 partial sums per point out
            roof["algorithmic_bytes_per_launch"] = 8.0 * (M * M + pts_per_launch * (M + M / 128.0))
        if world == 1 and not inner:
            progress["stage"] = "resident-points rate and points-style rate"
            # the same work through the C ABI with the points already resident in HBM: mik_factor + mik_predict per step
            # (round 1-2 headline; what execute() costs on top of it is host_overhead above)
            model._set_problem(h)
            P = model._prepare("grid", axes, None)
            P.load(h, ndim)
            h.factor(); h.predict(); h.synchronize()  # noqa: E702
            t1 = time.perf_counter()
            for _ in range(K):
                if kw:
                    h.predict_moving_window(kw)
                else:
                    h.factor()
                    h.predict()
            h.synchronize()
            d1 = time.perf_counter() - t1
            out["resident"] = {"value": npt_total * K / d1, "unit": "grid-points/s", "ms_per_step": d1 / K * 1e3,
                               "what": "mik_factor + mik_predict per step on points resident in HBM, results left in page-locked memory"}
            out["host_overhead"]["ms_per_step"] = dt / K * 1e3 - d1 / K * 1e3
            out["host_overhead"]["frac"] = 1.0 - d1 / dt
            if n_gpus == 1:
                # style='points': the same points handed over as host arrays (npt x d doubles over PCIe)
                parts = shard_points(cfg, 0, 1)
                best = None
                for _ in range(3):  # (the first call allocates the page-locked staging buffer)
                    t1 = time.perf_counter()
                    zz, sss = model.execute("points", *parts, **exe_kw)
                    d1 = time.perf_counter() - t1
                    best = d1 if best is None else min(best, d1)
                out["execute_points_style"] = {"value": npt_total / best, "unit": "grid-points/s",
                                               "includes": "H2D of npt x d point coordinates (page-locked staging) on top of everything the "
                                                           "grid call does"}
                zg, ssg = (np.asarray(a).ravel() for a in res)
                out["checksum"] = {"z_sum": float(zg.sum()), "ss_sum": float(ssg.sum()),
                                   "grid_vs_points_max_abs_dz": float(np.abs(zg - zz).max()), "grid_vs_points_max_abs_dss": float(np.abs(ssg - sss).max())}
        if n_gpus == 1 and world == 1 and not inner and not kw and args.config == 2 and not args.no_other:
            # BASELINE's other configurations and the moving window under the same clock (3 steps, 1 warm-up each; parity against
            # the stored reference slabs).  Keys are pinned by tests/test_bench_host.py.
            out["other_configs"] = {}
            for key, cno, win in (("config3", 3, None), ("config4", 4, None), ("config5", 5, None),
                                  ("moving_window_k10", 2, 10), ("moving_window_k100", 2, 100)):
                progress["stage"] = "other_configs: " + key
                try:
                    # (the moving-window calls are milliseconds long: 3 warm-up calls -- a new handle's landing zones come out of the page-locked pool only from
                    # its third call on, and hipHostMalloc after the large configs costs more than the whole k = 10 call -- and 10 timed ones)
                    out["other_configs"][key] = other_config_line(cno, win, steps=10, warmup=3) if win else other_config_line(cno, win)
                except Exception as e:  # noqa: BLE001
                    out["other_configs"][key] = {"value": None, "error": repr(e)[:300]}
        if n_gpus == 1 and not args.no_cpu and not inner:
            progress["stage"] = "cpu_baseline leg"
            try:
                cb, (cp, cz, css) = cpu_baseline(cfg, coords, values, args.cpu_sample, window=kw, full=args.cpu_protocol == "full")
                # parity of the GPU path on the very points the CPU baseline kriged
                model._set_problem(h)
                h.set_points(*[np.ascontiguousarray(cp[:, d]) for d in range(ndim)])
                if kw:
                    h.predict_moving_window(kw)
                else:
                    h.factor()
                    h.predict()
                gz, gss = h.get_results()
                cb["gpu_vs_cpu_max_abs_dz"] = float(np.abs(gz - cz).max())
                cb["gpu_vs_cpu_max_abs_dss"] = float(np.abs(gss - css).max())
                out["cpu_baseline"] = cb
            except Exception as e:  # the bench line must still come out
                out["cpu_baseline"] = {"value": None, "unit": "grid-points/s", "cores": None, "kind": "port",
                                       "sample": "failed: %r" % (e,)}
        if n_gpus == 1 and world == 1 and not inner and not kw and args.config == 2 and not args.no_cpu and args.full_grid_budget > 0:
            # parity over the WHOLE headline grid: one execute('grid') of all 10^6 points against the reference kriging the same grid
            # slab by slab for --full-grid-budget seconds (spread over the grid; `bench.py --full-parity` = without the bound)
            progress["stage"] = "whole-grid parity against the reference"
            try:
                out["full_grid_parity"] = full_grid_parity(2, budget_s=args.full_grid_budget)
            except Exception as e:  # noqa: BLE001
                out["full_grid_parity"] = {"ok": None, "error": repr(e)[:300]}
        config.update(scalars_for_driver(out))  # what the driver's record keeps: flat scalar keys of `config`
        emit(json.dumps(out))
    elif world > 1:
        pg.all_gather_object({"rank": rank, "device": int(os.environ["MIK_DEVICE"]), "predict_ms": last.get("predict_ms")})
    progress["done"] = True
    if pg is not None:
        pg.barrier()  # nobody tears its communicator down while another rank is still inside a collective
    if executor is not None:
        executor.close()
    h.close()


if __name__ == "__main__":
    main()
