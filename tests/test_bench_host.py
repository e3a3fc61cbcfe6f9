"""Host side of bench.py without a GPU: the workloads are the BASELINE configs, the weak-scaling shards tile the grid in
the reference's meshgrid order, the synthetic data are deterministic, and the JSON metric is BASELINE.json's."""
import json
import os
import re
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_configs_are_the_baseline_ones():
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    txt = base["configs"]
    c2, c3, c4 = bench.CONFIGS[2], bench.CONFIGS[3], bench.CONFIGS[4]
    assert "N=5000" in txt[1] and "1000" in txt[1] and "exponential" in txt[1]
    assert (c2["n"], c2["grid"], c2["model"], c2["seed"]) == (5000, (1000, 1000), "exponential", 2)
    assert "N=2000" in txt[2] and "gaussian" in txt[2]
    assert (c3["n"], c3["grid"], c3["model"], c3["ndim"]) == (2000, (200, 200, 50), "gaussian", 3)
    assert "regional_linear" in txt[3]
    assert c4["rl"] and len(c4["wells"]) == 3 and c4["n"] == 4000 and c4["grid"] == (1024, 1024)
    assert bench.CONFIGS[5]["n"] == 8000 and bench.CONFIGS[5]["grid"][0] == 4096
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "kriged grid-points/sec (z + sigma^2), OK2D N=5000 on 1000x1000 grid" in src
    assert re.sub("[²×]", "", base["metric"]).split(",")[0].startswith("kriged grid-points/sec")


def test_synthetic_inputs_are_deterministic_and_as_specified():
    (x, y), v = bench.synth(2, 5000, 2)
    (x2, y2), v2 = bench.synth(2, 5000, 2)
    assert np.array_equal(x, x2) and np.array_equal(v, v2)
    rng = np.random.default_rng(2)  # x, y = rng.random(N) each; v = sin(6x) cos(4y) + 0.1 N(0,1)
    xr, yr = rng.random(5000), rng.random(5000)
    assert np.array_equal(x, xr) and np.array_equal(y, yr)
    assert np.allclose(v, np.sin(6 * xr) * np.cos(4 * yr) + 0.1 * rng.standard_normal(5000))
    assert bench.internal_params("exponential", [1.0, 0.3, 0.1]) == [0.9, 0.3, 0.1]
    assert bench.internal_params("linear", [2.0, 0.1]) == [2.0, 0.1]


def test_weak_scaling_shards_tile_the_enlarged_grid():
    cfg = bench.CONFIGS[2]
    world = 4
    shards = [bench.shard_points(cfg, r, world) for r in range(world)]
    assert all(s[0].size == 1000 * 1000 for s in shards)  # per-GPU work fixed: weak scaling
    gx, gy = np.linspace(0, 1, 1000), np.linspace(0, 1, 1000 * world)
    X, Y = np.meshgrid(gx, gy)
    assert np.array_equal(np.concatenate([s[0] for s in shards]), X.ravel())
    assert np.array_equal(np.concatenate([s[1] for s in shards]), Y.ravel())
    c3 = bench.CONFIGS[3]
    s3 = [bench.shard_points(c3, r, 2) for r in range(2)]
    assert all(len(s) == 3 and s[0].size == 200 * 200 * 50 for s in s3)
    assert s3[0][2].max() < s3[1][2].min()  # z slabs stack


def test_bench_refuses_to_run_without_a_gpu():
    """No CPU fallback: on a box without a HIP device bench.py dies with the library's message and prints no JSON line."""
    try:
        from pykrige_amd import _lib

        if _lib.load().mik_device_count() > 0:
            return  # a GPU box: covered by the real bench run
    except Exception:
        pass
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--no-cpu"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "no HIP device" in r.stderr and r.stdout.strip() == ""


def test_cpu_protocol_slabs_are_rows_of_the_configs_own_grid():
    """the reference is timed on ROW SLABS of the grid (style='grid' with a y-subrange), not on random points."""
    for c, full_rows in ((2, 1000), (4, 1024), (5, 4096)):
        cfg = bench.CONFIGS[c]
        pts = bench.row_slab(cfg, 16384)
        nx = cfg["grid"][0]
        assert pts.shape[1] == 2 and pts.shape[0] >= 16384 and pts.shape[0] % nx == 0
        gx, gy = np.linspace(0, 1, nx), np.linspace(0, 1, full_rows)
        assert np.array_equal(pts[:nx, 0], gx)                      # whole rows, x fastest (the reference's meshgrid order)
        assert np.all(np.isin(np.unique(pts[:, 1]), gy))            # y values are rows of the full grid
    p3 = bench.row_slab(bench.CONFIGS[3], 4096)
    assert p3.shape[1] == 3 and p3.shape[0] >= 4096 and np.unique(p3[:, 2]).size == 1
    d = bench.host_description()
    assert d["host_cpus"] >= 1 and "numpy" in d


def test_exchange_names_match_the_header():
    hdr = open(os.path.join(ROOT, "include", "mikrige.h")).read()
    assert "0 = none (one device), 1 = RCCL broadcast, 2 = peer copies (scatter + all-gather), 3 = every device factored" in hdr
    assert bench.EXCHANGE_CODES == {"auto": 0, "rccl": 1, "peer": 2, "redundant": 3}
    assert bench.EXCHANGE_NAMES[1] == "rccl_bcast" and bench.EXCHANGE_NAMES[2].startswith("peer")


def test_live_traffic_collection_degrades_to_a_reason(monkeypatch):
    monkeypatch.setattr(bench.shutil, "which", lambda name: None)
    res, why = bench.collect_traffic_live(["--steps", "1"], "void mik::k_contract")
    assert res is None and "rocprofv3" in why


def test_tracked_pmc_summary_contains_the_dominant_kernel():
    """profiles/k_contract_traffic.json (bench.py's fall-back for roofline.traffic) must cite a TRACKED per-kernel PMC summary that
    really holds the dominant kernel's counters -- round 2's summary had lost its k_contract rows and nobody noticed."""
    tj = json.load(open(os.path.join(ROOT, "profiles", "k_contract_traffic.json")))
    src = tj["source"].split(" ")[0]
    assert src.startswith("profiles/") and os.path.exists(os.path.join(ROOT, src)), src
    rows = {}
    for line in list(open(os.path.join(ROOT, src)))[1:]:
        f = line.rstrip("\n").rsplit(",", 4)  # kernel names contain commas: the numeric fields are the last four
        rows.setdefault(f[0], {})[f[1]] = float(f[3])
    kern = "void mik::" + tj["kernel"]
    assert kern in rows, "the cited PMC summary has no rows for %s" % kern
    for ctr in ("FETCH_SIZE", "WRITE_SIZE", "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "TCC_HIT_sum", "TCC_MISS_sum"):
        assert rows[kern].get(ctr, 0.0) > 0.0, "%s: no %s" % (kern, ctr)
    assert any(k.startswith("void mik::k_rhs<") for k in rows) and "mik::k_ss_reduce" in rows
    c = rows[kern]
    assert abs(tj["hbm_bytes_per_launch"] - (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0) <= 1e-6 * tj["hbm_bytes_per_launch"]
    assert abs(tj["tcc_hit_rate"] - c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])) < 1e-9
    busy = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
    assert 0.5 < busy < 1.0 and abs(busy - tj["mfma_busy_fraction"]) < 1e-9
    # and the bench line of the same evidence run agrees with it to a few per cent (live collection in the same run)
    b = json.load(open(os.path.join(ROOT, src.replace("_rocprofv3_pmc_per_kernel.csv", ".json"))))
    assert b["roofline"]["kernel"] == "k_contract" and b["roofline"]["traffic"] is not None
    assert abs(b["roofline"]["traffic"] - tj["hbm_bytes_per_launch"]) <= 0.05 * tj["hbm_bytes_per_launch"]


def test_other_configs_key_set_and_trial_flags_are_pinned():
    """Round 4: the default N = 1 run also times BASELINE configs 3, 4, 5 and the moving window (k = 10, 100) and checks each against
    the stored reference slab; the line's key set is what the driver's BENCH record will show."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    for key in ("config3", "config4", "config5", "moving_window_k10", "moving_window_k100"):
        assert '("%s",' % key in src
    for field in ('"value"', '"ms_per_step"', '"roofline"', '"max_abs_dz"', '"max_abs_dss"', '"parity_source"'):
        assert field in src[src.index("def other_config_line"):src.index("MW_KERNELS = {")]
    assert 'out["other_configs"]' in src
    # round 6: the driver's record keeps only SCALAR keys of `config`: everything it must hold is flattened there (scalars_for_driver)
    assert "config.update(scalars_for_driver(out))" in src
    line = {"value": 3.0e7, "ms_per_step": 500.0, "roofline": {"frac": 0.78, "kernel": "k_contract_spg"}, "max_abs_dz": 1e-11, "max_abs_dss": 2e-11,
            "phases_ms_per_step": {"invert": 10.0}, "parity_source": "x"}
    assert bench.compact_other(line) == {"value": 3.0e7, "ms_per_step": 500.0, "frac": 0.78, "kernel": "k_contract_spg", "max_abs_dz": 1e-11,
                                         "max_abs_dss": 2e-11, "invert_ms": 10.0}
    assert bench.compact_other({"value": None, "error": "boom"}) == {"value": None, "ms_per_step": None, "frac": None, "kernel": None,
                                                                     "max_abs_dz": None, "max_abs_dss": None, "error": "boom"}
    mg = bench.compact_multi_gpu({"ranks": [{"rank": 0, "predict_ms": 5.0}, {"rank": 1, "predict_ms": 6.0}], "exchange_path": "rccl_bcast", "rccl_ranks": 2})
    assert mg["rccl_ranks"] == 2 and mg["exchange_path"] == "rccl_bcast" and mg["per_device_predict_ms"] == [5.0, 6.0]
    mg = bench.compact_multi_gpu({"per_device_predict_ms": [1.0, 2.0], "exchange_path": "peer_scatter_allgather", "rccl_ranks": 0, "exchange_ms": 3.0})
    assert set(mg) == {"rccl_ranks", "exchange_path", "per_device_predict_ms", "exchange_ms", "exchange_wait_ms", "exchange_bytes", "exchange_fallbacks",
                       "exchange_note"}
    for flag in ("--no-trials", "--pretrial-budget", "--no-other", "--sparse", "--sparse-rows", "--sort-points", "--sparse-lanes", "MIK_BENCH_TRIALS"):
        assert flag in src
    # the stored slabs those checks read exist and are the configs' own
    for cno, n in ((3, 2000), (4, 4000), (5, 8000)):
        g = bench.golden_slab(cno)
        assert g["x"].size == n == bench.CONFIGS[cno]["n"] and g["z"].size >= 16384
        assert str(g["model"]) == bench.CONFIGS[cno]["model"] and g["params_user"].tolist() == bench.CONFIGS[cno]["params"]
    mw = np.load(os.path.join(ROOT, "tests", "golden", "fullsize", "mw_c2.npz"))
    assert mw["x"].size == bench.CONFIGS[2]["n"] and mw["windows"].tolist() == [10, 100]


def test_driver_record_keys_are_flat_scalars_and_pinned():
    """Round 6: what the driver's BENCH record keeps of the line is the scalar keys of `config`.  A fabricated line of the default run
    (other configs, CPU leg, whole-grid parity) and of an 8-GPU run must flatten into scalars only, under these names."""
    other = {"value": 3.0e7, "ms_per_step": 56.0, "roofline": {"frac": 0.78, "kernel": "k_contract_spg"}, "max_abs_dz": 1e-11, "max_abs_dss": 2e-11,
             "phases_ms_per_step": {"invert": 13.0}}
    out = {"phases_ms_per_step": {"assemble": 0.1, "invert": 4.3, "exchange": 0.0, "exchange_not_overlapped": 0.0, "rhs": 73.0, "contract": 356.0,
                                  "predict_total": 365.0},
           "other_configs": {k: dict(other) for k in bench.SHORT},
           "cpu_baseline": {"value": 1466.0, "kind": "reference", "cores": 16, "cond_1": 4.6e6, "gpu_vs_cpu_max_abs_dz": 1e-10,
                            "gpu_vs_cpu_max_abs_dss": 3e-10, "slab_points": 17000, "host": {"nested": 1}},
           "full_grid_parity": {"points_checked": 10 ** 6, "points_total": 10 ** 6, "coverage": 1.0, "max_abs_dz": 1e-10, "max_abs_dss": 3e-10,
                                "cond_1": 4.6e6, "reference_points_per_s": 8000.0, "ok": True, "worst_dz_at": [0.1, 0.2]},
           "roofline": {"kernel": "k_contract", "avg_launch_ms": 44.5, "traffic": 1.1e11, "algorithmic_bytes_per_launch": 5.2e9, "note": "x"}}
    c = bench.scalars_for_driver(out)
    assert all(v is None or isinstance(v, (int, float, str, bool)) for v in c.values()), c
    for key in ("phase_invert_ms", "phase_contract_ms", "phase_predict_total_ms", "c3_value", "c3_frac", "c4_value", "c5_value", "c5_frac", "c5_kernel",
                "c5_max_abs_dz", "c5_max_abs_dss", "c5_invert_ms", "mw_k10_value", "mw_k100_value", "mw_k100_frac", "cpu_value", "cpu_kind", "cpu_cores",
                "cpu_cond_1", "gpu_vs_cpu_max_abs_dz", "gpu_vs_cpu_max_abs_dss", "c2_fullgrid_points_checked", "c2_fullgrid_coverage",
                "c2_fullgrid_max_abs_dz", "c2_fullgrid_max_abs_dss", "c2_fullgrid_ok", "roofline_kernel", "roofline_avg_launch_ms", "roofline_traffic"):
        assert key in c, key
    mg = {"multi_gpu": {"per_device_predict_ms": [41.0, 42.5, None], "exchange_path": "rccl_bcast", "rccl_ranks": 8, "exchange_ms": 6.0,
                        "exchange_wait_ms": 1.0, "exchange_bytes": 2.6e8},
          "factor_exchange_trial": {"budget_s": 60.0, "rccl_exchange_ms": 6.1, "skipped_for_budget": ["overlap"]}}
    c = bench.scalars_for_driver(mg)
    assert all(v is None or isinstance(v, (int, float, str, bool)) for v in c.values()), c
    assert c["rccl_ranks"] == 8 and c["exchange_path"] == "rccl_bcast" and c["exchange_bytes"] == 2.6e8
    assert c["predict_ms_slowest_device"] == 42.5 and c["per_device_predict_ms"] == "41.00,42.50"
    assert c["trial_rccl_exchange_ms"] == 6.1 and c["trial_skipped_for_budget"] == '["overlap"]'
    # and nothing nested is put under `config` by main() any more
    src = open(os.path.join(ROOT, "bench.py")).read()
    for gone in ('config["other_configs"]', 'config["multi_gpu"]', 'config["phases_ms_per_step"]', '"factor_exchange_trial": trial'):
        assert gone not in src, gone


def test_whole_grid_parity_plan_covers_every_point_once_and_spreads():
    """oracle/full_grid.py (round 6): slabs are whole rows / z planes that tile the grid exactly; the visiting order spreads any prefix."""
    from oracle import full_grid as fg

    for cno in (2, 3, 4, 5):
        axes = bench.full_grid_axes(cno)
        plan = fg.slab_plan(axes, bench.FULL_GRID[cno]["target"])
        seen = np.zeros(axes[-1].size, dtype=int)
        for ax, sl in plan:
            assert all(np.array_equal(a, b) for a, b in zip(ax[:-1], axes[:-1])) and np.array_equal(ax[-1], axes[-1][sl])
            seen[sl] += 1
            assert int(np.prod([a.size for a in ax])) <= max(bench.FULL_GRID[cno]["target"], int(np.prod([a.size for a in axes[:-1]])))
        assert (seen == 1).all()
    assert [a.size for a in bench.full_grid_axes(5)] == [4096, 64] and bench.FULL_GRID[5]["strip"][0] < 512 < bench.FULL_GRID[5]["strip"][1]
    o = fg.spread_order(20)
    assert sorted(o) == list(range(20)) and o[:4] == [0, 16, 8, 4]

    class Fake:  # a "reference" that returns f(x, y) on the slab: compare() must index the whole-grid arrays the same way
        def execute(self, style, gx, gy, backend):
            X, Y = np.meshgrid(gx, gy)
            return X + 10 * Y, X - Y

    gx, gy = np.linspace(0, 1, 7), np.linspace(0, 1, 11)
    X, Y = np.meshgrid(gx, gy)
    zz, ss = X + 10 * Y, X - Y
    zz[6, 3] += 1e-3
    r = fg.compare(Fake(), zz, ss, [gx, gy], 14)
    assert r["points_checked"] == 77 and r["coverage"] == 1.0 and abs(r["max_abs_dz"] - 1e-3) < 1e-12 and r["max_abs_dss"] == 0.0
    assert np.allclose(r["worst_dz_at"], [gx[3], gy[6]])
    r = fg.compare(Fake(), zz, ss, [gx, gy], 14, budget_s=0.0)
    assert r["slabs_checked"] == 1 and 0 < r["coverage"] < 1


def test_whole_grid_checker_runs_several_reference_processes():
    """oracle/full_grid.py: W reference processes side by side (spawned; each builds bench.reference_model from the recipe, 8 BLAS threads through its
    environment) return the slabs of the one-process run to the rounding of the reference's own BLAS thread count; a budget of zero still checks one
    slab per process."""
    import pytest

    from oracle import full_grid as fg
    from oracle import ref_package as rp

    if not rp.available():
        pytest.skip("oracle/_ref/pykrige_py.zip not staged")
    pk = rp.import_reference(stub_statistics=True)
    cfg = dict(bench.CONFIGS[4], n=200)
    coords, values = bench.synth(cfg["seed"], cfg["n"], 2)
    rm = bench.reference_model(pk, cfg, coords, values)
    axes = [np.linspace(0, 1, 40), np.linspace(0, 1, 21)]
    z, ss = rm.execute("grid", *axes, backend="vectorized")
    one = fg.compare(rm, z, ss, axes, 160)
    two = fg.compare(rm, z, ss, axes, 160, recipe=(cfg, coords, values), workers=2)
    assert one["points_checked"] == two["points_checked"] == 840 and one["reference_processes"] == 1 and two["reference_processes"] == 2
    assert one["max_abs_dz"] <= 1e-11 and two["max_abs_dz"] <= 1e-11 and two["max_abs_dss"] <= 1e-11  # (the reference's own sums move in the 14th digit with slab shape and thread count)
    few = fg.compare(rm, z, ss, axes, 160, recipe=(cfg, coords, values), workers=2, budget_s=0.0)
    assert few["slabs_checked"] == 2 and 0 < few["coverage"] < 1


def test_cpu_leg_times_the_reference_itself_where_it_is_staged():
    """cpu_baseline on a tiny stand-in config: with oracle/_ref/pykrige_py.zip staged (oracle/build_ref.sh) the vectorized leg is the
    reference's own execute() -- kind "reference" -- and agrees with the port; the slab axes are the slab's points."""
    from oracle import ref_package as rp

    if not rp.available():
        import pytest

        pytest.skip("oracle/_ref/pykrige_py.zip not staged")
    for cno, extra in ((3, {}), (4, {})):
        cfg = dict(bench.CONFIGS[cno], n=120, grid=tuple(min(g, 24) for g in bench.CONFIGS[cno]["grid"]), **extra)
        coords, values = bench.synth(cfg["seed"], cfg["n"], cfg["ndim"])
        out, (pts, z, ss) = bench.cpu_baseline(cfg, coords, values, 96, full=False)
        assert out["kind"] == "reference" and out["vectorized"]["kind"] == "reference", out
        assert out["vectorized"]["port_vs_reference_max_abs_dz"] < 1e-9 and out["vectorized"]["port_vs_reference_max_abs_dss"] < 1e-9
        ax = bench.row_slab_axes(cfg, 96)
        assert pts.shape[0] == int(np.prod([a.size for a in ax])) == z.size
