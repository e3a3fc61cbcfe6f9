#!/usr/bin/env python
"""bench.py -- kriged grid-points/sec (z + sigma^2) of the HIP execute() path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 2|3|4|5] [--no-cpu]

One "step" = one pass of the hot path over the workload with inputs already resident in HBM:
kriging-matrix assembly + inverse (K1, K2) [+ RCCL broadcast of the inverse when N > 1] + RHS assembly
and contraction (K3) for every grid point of this rank's shard; z and sigma^2 stay in HBM.
Default workload = BASELINE.json configs[1]: OrdinaryKriging 2D, N=5000 stations, 1000x1000 grid,
exponential variogram [1.0, 0.3, 0.0], fp64, synthetic stations (seed 2).
N > 1 (launched by torch.distributed.run, one rank per GPU): WEAK scaling -- every rank kriges its own
1000x1000 slab of a 1000 x (1000 N) grid against the same stations; rank 0 assembles + inverts and the
inverse is broadcast over RCCL/xGMI by the library (mik_bcast_factor); no other collective.
The host-side rendezvous / barrier / max-over-ranks of the time go through pykrige_amd.dist.SocketGroup (TCP, from the
launcher's RANK/WORLD_SIZE/MASTER_* environment); torch is not imported unless that fails (then: its gloo group).
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6  # MI355X FP64 matrix peak (public spec; = 256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz).
# The local microarch guide lists no FP64 row; tools/ubench_f64.hip measures the achievable rate (profiles/).

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


def cpu_baseline(cfg, coords, values, sample_pts):
    """The reference's own backend='C' native loop (oracle/_ref, compiled from /root/reference's
    lib/cok.pyx) -- or, if that build is absent, the numpy oracle -- on a bounded sample of the workload."""
    from oracle import kriging_oracle as ko
    from oracle import ref_c_loop as rc

    ndim = cfg["ndim"]
    st = ko.KrigingState(ndim=ndim, coords_orig=np.stack(coords, 1), values=values, model=cfg["model"],
                         params=internal_params(cfg["model"], cfg["params"]), scaling=[1.0] * (ndim - 1),
                         angle=[0.0] * (2 * ndim - 3), regional_linear=bool(cfg.get("rl")),
                         point_log=np.array(cfg["wells"]) if cfg.get("wells") else None)
    rng = np.random.default_rng(99)
    pts = rng.random((sample_pts, ndim))
    try:
        import threadpoolctl

        cores = max([p.get("num_threads", 1) for p in threadpoolctl.threadpool_info()] or [1])
    except Exception:
        cores = os.cpu_count() or 1
    use_c = rc.available() and ndim == 2 and not cfg.get("rl") and cfg["model"] != "hole-effect"
    # fixed costs timed on their own so that the steady-state (per-point) rates can be separated
    import scipy.linalg

    t0 = time.perf_counter()
    a = ko.kriging_matrix(st)
    t_mat = time.perf_counter() - t0
    t0 = time.perf_counter()
    a_inv = scipy.linalg.inv(a)
    t_inv = time.perf_counter() - t0
    t0 = time.perf_counter()
    zv, ssv = ko.solve_points(st, pts, a_inv=a_inv)
    t_vec = time.perf_counter() - t0
    npt_full = int(np.prod(cfg["grid"]))
    vec = {"value": sample_pts / (t_mat + t_inv + t_vec), "steady_state": sample_pts / t_vec, "kind": "port",
           "what": "numpy/scipy restatement of backend='vectorized' (oracle/kriging_oracle.py), 4096-point slabs",
           "full_grid_projection": npt_full / (t_mat + t_inv + npt_full * t_vec / sample_pts)}
    if use_c:
        t0 = time.perf_counter()
        z, ss, t_loop = rc.c_backend(st, pts)  # its native loop runs scipy.linalg.inv itself (cok.pyx:53)
        dt = time.perf_counter() - t0
        kind = "reference"
        what = "PyKrige lib/cok.pyx _c_exec_loop (backend='C') incl. its scipy.linalg.inv"
        # the separately timed inverse only separates cleanly when the loop dominates; otherwise report no split
        per_pt = (t_loop - t_inv) / sample_pts if t_loop > 1.5 * t_inv else None
    else:
        z, ss, dt = zv, ssv, t_mat + t_inv + t_vec
        kind = "port"
        what = vec["what"]
        per_pt = t_vec / sample_pts
    return dict(value=sample_pts / dt, unit="grid-points/s", cores=int(cores), kind=kind,
                sample="%d random points of the same workload, %s, matrix assembly + inverse + loop = %.1f s "
                       "(fixed costs included)" % (sample_pts, what, dt),
                steady_state=None if per_pt is None else 1.0 / per_pt,
                full_grid_projection=None if per_pt is None else npt_full / (t_mat + t_inv + npt_full * per_pt),
                fixed_costs_s={"matrix": t_mat, "inverse": t_inv}, vectorized=vec,
                c_vs_vectorized_max_abs_dz=float(np.abs(z - zv).max()), c_vs_vectorized_max_abs_dss=float(np.abs(ss - ssv).max()),
                host_cpus=os.cpu_count()), (pts, z, ss)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-sample", type=int, default=3072)
    ap.add_argument("--symmetric", type=int, default=None)
    ap.add_argument("--chunk", type=int, default=None)
    ap.add_argument("--engine", choices=["mfma", "valu"], default=None)
    ap.add_argument("--factor", choices=["auto", "sweep", "lu"], default=None, help="force the inverse path")
    ap.add_argument("--moving-window", type=int, default=None, metavar="K",
                    help="time moving-window kriging (n_closest_points=K) on the same workload instead (not the headline metric)")
    args = ap.parse_args()

    # Keep stdout clean for the ONE JSON line: gloo / RCCL print banners from C++ to fd 1, so fd 1 points at
    # stderr while the benchmark runs and is restored just before the JSON is printed.
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)

    def emit(line):
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        print(line, flush=True)
        os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
        args.gpus = world
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
    pts = shard_points(cfg, rank, world)
    npt = pts[0].size

    ndev = _lib.load().mik_device_count()
    wells = np.array(cfg["wells"]) if cfg.get("wells") else None

    def make_handle():
        hh = _lib.Handle(local_rank % max(ndev, 1))  # one GPU per rank on a real node; wraps only on a 1-GPU test box
        if args.symmetric is not None:
            hh.set_option("symmetric", args.symmetric)
        if args.chunk is not None:
            hh.set_option("chunk", args.chunk)
        if args.engine is not None:
            hh.set_option("engine", 1 if args.engine == "valu" else 0)
        if args.factor is not None:
            hh.set_option("factor", {"auto": 0, "sweep": 1, "lu": 2}[args.factor])
        hh.set_problem(ndim=ndim, xs=coords[0], ys=coords[1], zs=coords[2] if ndim == 3 else None, values=values,
                       model_id=_lib.MODEL_IDS[cfg["model"]], params=internal_params(cfg["model"], cfg["params"]),
                       regional_linear=bool(cfg.get("rl")), wells=wells)
        hh.set_points(pts[0], pts[1], pts[2] if ndim == 3 else None)
        return hh

    h = make_handle()

    # How the factored matrix reaches every rank is decided by measurement, outside the timed region: (A) rank 0 factors
    # and the library broadcasts T and c over RCCL/xGMI, or (B) every rank factors for itself (no collective at all).
    # Ranks != 0 wait for rank 0's factorisation in (A) anyway, so (A) wins only if the broadcast beats nothing -- it
    # usually does not, and the trial says so in the JSON.  Every RCCL call of the trial runs under a watchdog: a wedged
    # bootstrap or collective degrades to (B) on a fresh handle instead of hanging the benchmark.
    exchange, trial, leaked = "none", None, False
    if world > 1 and not args.moving_window:
        mode = os.environ.get("MIK_BENCH_EXCHANGE", "auto")  # auto | rccl | redundant
        if mode == "redundant":
            exchange = "redundant_factor"
        else:
            import threading

            from pykrige_amd.dist import init_rccl

            exchange = init_rccl(h, pg)  # "rccl_bcast", or "redundant_factor (...)"
            if exchange == "rccl_bcast":
                def watchdog(fn, limit):
                    box = {}

                    def run():
                        try:
                            fn()
                            box["ok"] = True
                        except Exception as e:  # noqa: BLE001
                            box["err"] = repr(e)[:120]

                    th = threading.Thread(target=run, daemon=True)
                    t0 = time.perf_counter()
                    th.start()
                    th.join(limit)
                    return box.get("ok", False), time.perf_counter() - t0, box.get("err"), th.is_alive()

                def via_bcast():
                    if rank == 0:
                        h.factor()
                    h.bcast_factor(0)

                limit = float(os.environ.get("MIK_RCCL_BCAST_TIMEOUT", "60"))
                ta = tb = None
                for _ in range(2):  # first round pays RCCL's lazy channel set-up and the buffer allocations
                    pg.barrier()
                    ok, ta, err, hung = watchdog(via_bcast, limit)
                    res = pg.all_gather_object((ok, ta, err))
                    if not all(r[0] for r in res):
                        why = next((r[2] for r in res if r[2]), "broadcast did not finish within %.0f s" % limit)
                        exchange = "redundant_factor (rccl broadcast failed: %s)" % why
                        if hung:  # this rank's stream is stuck behind the collective: abandon the handle
                            leaked = True
                            h = make_handle()
                        break
                    ta = max(r[1] for r in res)
                    pg.barrier()
                    t0 = time.perf_counter()
                    h.factor()
                    tb = pg.all_reduce_max(time.perf_counter() - t0)
                if exchange == "rccl_bcast":
                    trial = {"rccl_bcast_ms": ta * 1e3, "redundant_factor_ms": tb * 1e3}
                    if mode != "rccl" and tb <= ta:
                        exchange = "redundant_factor (measured faster than rank-0 factor + RCCL broadcast)"

    def sync():
        h.synchronize()  # device idle (mik_predict / mik_bcast_factor already block; this is the explicit bracket)
        if pg is not None:
            pg.barrier()

    tsum = dict(assemble_ms=0.0, invert_ms=0.0, rhs_ms=0.0, contract_ms=0.0, predict_ms=0.0, contract_launches=0,
                contract_flops_executed=0.0)

    def step(record):
        if args.moving_window:
            h.predict_moving_window(args.moving_window)
            if record:
                tsum["predict_ms"] += h.timing()["predict_ms"]
            return
        if exchange == "rccl_bcast":
            if rank == 0:
                h.factor()
            h.bcast_factor(0)
        else:
            h.factor()
        if record and (rank == 0 or exchange != "rccl_bcast"):
            t = h.timing()
            tsum["assemble_ms"] += t["assemble_ms"]
            tsum["invert_ms"] += t["invert_ms"]
        h.predict()  # blocking: returns after the stream has drained
        if record:
            t = h.timing()
            for k in ("rhs_ms", "contract_ms", "predict_ms", "contract_launches", "contract_flops_executed"):
                tsum[k] += t[k]
            tsum["factor_path"], tsum["symmetric"], tsum["engine"] = t["factor_path"], t["symmetric"], t["engine"]

    for _ in range(args.warmup):
        step(False)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    sync()
    dt = time.perf_counter() - t0
    if pg is not None:
        dt = pg.all_reduce_max(dt)

    if rank == 0:
        K = args.steps
        M = cfg["n"] + (ndim if cfg.get("rl") else 0) + (len(cfg["wells"]) if cfg.get("wells") else 0) + 1
        total_pts = npt * world
        value = total_pts * K / dt
        launches = max(1, int(tsum["contract_launches"]))
        avg_launch_s = tsum["contract_ms"] * 1e-3 / launches
        pts_per_launch = npt * K / launches
        algo_flops_per_launch = 2.0 * M * M * pts_per_launch  # 2 M^2 per point for w = A_inv . b
        achieved = algo_flops_per_launch / avg_launch_s / 1e12 if avg_launch_s > 0 else 0.0
        executed = tsum["contract_flops_executed"] / (tsum["contract_ms"] * 1e-3) / 1e12 if tsum["contract_ms"] > 0 else 0.0
        traffic, traffic_note = None, None
        try:  # HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (not collected live)
            tj = json.load(open(os.path.join(ROOT, "profiles", "k_contract_traffic.json")))
            if tj["workload"] == cfg["name"] and not tsum.get("engine") and tsum.get("symmetric"):
                traffic = tj["hbm_bytes_per_launch"] * (pts_per_launch / tj["points_per_launch"])
                traffic_note = tj["source"]
        except Exception:
            pass
        if args.moving_window:
            emit(json.dumps({"metric": "kriged grid-points/sec (z + sigma^2), moving window n_closest_points=%d, %s"
                                        % (args.moving_window, cfg["name"]), "value": value, "unit": "grid-points/s",
                              "n_gpus": world, "steps": K, "warmup": args.warmup, "ms_per_step": dt / K * 1e3,
                              "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
                              "data": "synthetic", "config": {"workload": cfg["name"], "n_closest_points": args.moving_window}}))
            if pg is not None:
                pg.barrier()
            h.close()
            return
        out = {
            "metric": "kriged grid-points/sec (z + sigma^2), OK2D N=5000 on 1000x1000 grid" if args.config == 2
            else "kriged grid-points/sec (z + sigma^2), " + cfg["name"],
            "value": value, "unit": "grid-points/s", "n_gpus": world, "steps": K, "warmup": args.warmup,
            "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": cfg["name"], "stations": cfg["n"], "matrix_order": M,
                       "grid_points_per_gpu": npt, "grid_points_total": total_pts, "variogram": cfg["model"],
                       "variogram_parameters": cfg["params"], "factor_exchange": exchange, "factor_exchange_trial": trial,
                       "factor_path": {1: "spd-shift block sweep", 2: "pivoted block gauss-jordan", 3: "caller-supplied inverse", 4: "device pseudo-inverse"}.get(
                           tsum.get("factor_path"), "?"),
                       "symmetric_contraction": bool(tsum.get("symmetric"))},
            "roofline": {"bound": "mfma", "kernel": "k_contract_valu" if tsum.get("engine") else "k_contract", "achieved": achieved, "peak": FP64_MFMA_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved / FP64_MFMA_PEAK_TFLOPS, "traffic": traffic,
                         "traffic_unit": "bytes per launch (HBM-side, PMC)", "traffic_source": traffic_note,
                         "executed_tflops": executed, "frac_executed": executed / FP64_MFMA_PEAK_TFLOPS,
                         "note": "achieved / frac use the reference's 2 M^2 flops per point (SURVEY 8d); the kernel executes ~M^2 (symmetric half product), see executed_tflops / frac_executed",
                         "avg_launch_ms": avg_launch_s * 1e3,
                         "launches_per_step": launches / K, "algorithmic_flops_per_point": 2.0 * M * M},
            "phases_ms_per_step": {"assemble": tsum["assemble_ms"] / K, "invert": tsum["invert_ms"] / K,
                                   "rhs": tsum["rhs_ms"] / K, "contract": tsum["contract_ms"] / K,
                                   "predict_total": tsum["predict_ms"] / K},
        }
        if world == 1:  # the same step with host buffers handed over and results copied back (never `value`)
            t1 = time.perf_counter()
            h.set_points(pts[0], pts[1], pts[2] if ndim == 3 else None)
            h.factor()
            h.predict()
            zz, sss = h.get_results()
            out["pcie_inclusive"] = {"value": npt / (time.perf_counter() - t1), "unit": "grid-points/s",
                                     "includes": "H2D of the point coordinates, assemble+invert, predict, D2H of z and sigma^2"}
            out["checksum"] = {"z_sum": float(zz.sum()), "ss_sum": float(sss.sum())}
        if world == 1 and not args.no_cpu:
            try:
                cb, (cp, cz, css) = cpu_baseline(cfg, coords, values, args.cpu_sample)
                # parity of the GPU path on the very points the CPU baseline kriged
                h.set_points(*[cp[:, d] for d in range(ndim)])
                h.predict()
                gz, gss = h.get_results()
                cb["gpu_vs_cpu_max_abs_dz"] = float(np.abs(gz - cz).max())
                cb["gpu_vs_cpu_max_abs_dss"] = float(np.abs(gss - css).max())
                out["cpu_baseline"] = cb
            except Exception as e:  # the bench line must still come out
                out["cpu_baseline"] = {"value": None, "unit": "grid-points/s", "cores": None, "kind": "port",
                                       "sample": "failed: %r" % (e,)}
        emit(json.dumps(out))
    if pg is not None:
        pg.barrier()  # nobody tears its communicator down while another rank is still inside a collective
    if leaked:  # a handle was abandoned behind a stuck collective: skip every destructor
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)
    h.close()


if __name__ == "__main__":
    main()
