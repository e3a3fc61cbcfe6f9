"""CPU-side checks: the C-ABI library builds/loads and exports every entry point include/mikrige.h declares
(no compute call is made), the ctypes mirror agrees with the header, and the host-side logic
(parameter handling, anisotropy, bilinear external-Z lookup, argument validation) behaves like the
reference.  None of this needs a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

from oracle import kriging_oracle as ko

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    text = open(os.path.join(ROOT, "include", "mikrige.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mik_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from pykrige_amd import _lib, build

    build.build_library()  # no-op when up to date; cross-compiles for gfx950 without a GPU
    lib = _lib.load()
    names = _header_functions()
    assert len(names) >= 21
    for n in names:
        assert hasattr(lib, n), "libmikrige.so does not export %s" % n
    assert set(_lib.SIGNATURES) == set(names)


def test_library_size_stays_under_the_bar():
    """Ten translation units, code objects compressed in the fat binary (pykrige_amd/build.py).  The round-4 review's bar for the library (16.9 MB
    then) was 10 MB, the round-5 review's 2.2 MB (2.62 MB then); 1.94 MB at the end of round 6.  (A hipcc without --offload-compress builds it
    uncompressed and several times larger: that build is recorded in the .flags stamps and not held to the bar.)"""
    from pykrige_amd import build

    build.build_library()
    stamps = [open(os.path.join(build.OBJ, n + ".flags")).read() for n in build.UNITS if os.path.exists(os.path.join(build.OBJ, n + ".flags"))]
    if stamps and not all("--offload-compress" in st.split() for st in stamps):
        pytest.skip("this hipcc does not compress code objects")
    assert os.path.getsize(build.OUT) <= 2_200_000, os.path.getsize(build.OUT)


# Every kernel template the library may instantiate, with the DEFAULT code path or the documented fallback that launches it (round 6: "keep
# it pruned").  A kernel that is not in this table -- or an instantiation outside the pinned sets below -- fails the test: an experiment
# belongs in tools/ (tools/mik_k_experiments.h), not in libmikrige.so.
REACHABLE = {
    # K1 / set-up
    "k_assemble": "mik_factor: kriging matrix, 8 variogram ids x NDIM 1 (geographic) / 2 / 3", "k_geo_unit": "geographic stations -> unit vectors",
    "k_geo_unit_p": "geographic points -> unit vectors", "k_grid_points": "mik_set_grid: meshgrid + anisotropy on the device",
    "k_mask_count": "masked grids", "k_mask_scan": "masked grids", "k_mask_write": "masked grids", "k_checksum": "factor exchange: checksum of A^-1, c",
    "k_tri_pack": "factor exchange: pack / unpack the upper block triangle of A^-1",
    # K2 default sweep
    "k_diag_inv_b": "sweep: blocked diagonal-block inverse", "k_panel": "sweep: column panel (32 rows per block)", "k_update": "sweep: trailing update, 8 waves",
    "k_gemm128": "sweep: early-diagonal chain (two 128^3 products)", "k_gate": "sweep: gate hint", "k_copy_panel": "sweep: column panel copy (full sweep, pivoted)",
    "k_copy_panel_sym": "half sweep: column panel from the upper triangle", "k_mirror_upper": "half sweep: mirror", "k_symmetrize": "full sweep / pivoted: average the triangles",
    "k_add_diag": "corner fix after the shifted sweep", "k_cvec": "c = A^-1[:, :N] Z", "k_matvec": "probe of the inverse", "k_matvec3": "probe of the inverse",
    # K2 fallbacks: pivoted path (failed probe / non-SPD / custom variogram), pseudo-inverses (pseudo_inv=True)
    "k_piv_first": "pivoted fallback", "k_piv_step": "pivoted fallback", "k_swap_rows": "pivoted fallback", "k_swap_cols": "pivoted fallback",
    "k_transpose_rows": "pivoted fallback", "k_coo_add": "pseudo_inv: deflated inverse (duplicated stations)", "k_shift_diag": "pseudo_inv: null-space inverse",
    "k_lowrank_add": "pseudo_inv: null-space inverse", "k_set_identity": "pseudo_inv: Jacobi", "k_rownorm2": "pseudo_inv: Jacobi", "k_pinv_gemm": "pseudo_inv: Jacobi",
    "k_jac_step": "pseudo_inv: scalar Jacobi below 1536 rows", "k_bj_gram": "pseudo_inv: block Jacobi from 1536 rows", "k_bj_eig": "pseudo_inv: block Jacobi",
    "k_bj_rotate": "pseudo_inv: block Jacobi",
    # K3
    "k_rhs": "right-hand sides + z (8 variogram ids x NDIM; spherical: candidate tiles, 16- or 8-station flags)", "k_contract": "dense contraction (3 forms, see below)",
    "k_ss_reduce": "sigma^2 from the row-block partial sums", "k_contract_spg": "range-aware contraction, gathered row groups (default for spherical)",
    "k_contract_sp": "range-aware contraction, aligned blocks (matrix order > 23 168: 32-bit DMA offsets run out)", "k_ss_reduce_sp": "range-aware: sigma^2",
    "k_sp_cand": "range-aware: candidate tiles", "k_sp_lists_g": "range-aware: lists (gathered)", "k_sp_tiles_g": "range-aware: tile records (gathered)",
    "k_sp_lists": "range-aware: lists (aligned fallback)", "k_sp_tiles": "range-aware: tiles (aligned fallback)",
    "k_ps_keys": "points of a launch in Hilbert order", "k_ps_hist": "point sort", "k_ps_scan": "point sort", "k_ps_scatter": "point sort", "k_ps_gather": "point sort",
    "k_ps_bbox": "point sort", "k_ps_unsort": "point sort",
    # moving window
    "k_mw_knn": "neighbour search", "k_mw_knn_lane": "neighbour search, windows <= 16", "k_mw_knn_big": "neighbour search, large windows", "k_mw_pairdist": "custom variogram",
    "k_mw_geo_dist": "geographic windows", "k_mw_rhs": "moving-window right-hand sides (pivoting solvers)", "k_mw_rhs_table": "custom variogram",
    "k_mw_chol": "LDL^T in registers, 21 classes x (4 static models + dynamic)", "k_mw_chol_blocked": "windows > 256", "k_mw_solve": "pivoting fallback (hole-effect, custom, failed LDL^T)",
    "k_mw_solve_big": "pivoting fallback, large windows",
    # statistics / experimental variogram / self-tests
    "k_stat_dupes": "mik_statistics", "k_stat_matvec": "mik_statistics", "k_stat_reduce": "mik_statistics", "k_stat_update": "mik_statistics",
    "k_vg_minmax": "mik_experimental_variogram", "k_vg_bin": "mik_experimental_variogram",
    "k_selftest_mfma": "mik_selftest_mfma", "k_selftest_mfma4": "mik_selftest_mfma", "k_selftest_exp": "mik_selftest_exp",
}
PINNED = {  # families the round-5 review found carrying losers: exactly these instantiations
    "k_update": {"k_update<true, 2>", "k_update<false, 2>"},
    "k_panel": {"k_panel<1>"},
    "k_contract": {"k_contract<true, 2, true, false, true, false>",    # default: symmetric half product, triangular diagonal blocks
                   "k_contract<true, 2, true, false, false, false>",   # "tri" 0: whole diagonal blocks (cross-check of the parity tests)
                   "k_contract<false, 2, true, false, false, false>"},  # "symmetric" 0: the reference's w = A_inv b (cross-check)
    "k_contract_spg": {"k_contract_spg<2, true, true, false>", "k_contract_spg<2, true, true, true>"},  # default; MIK_SPG_PROF=1 diagnostic
    "k_contract_sp": {"k_contract_sp<2>"},
    "k_sp_tiles_g": {"k_sp_tiles_g<true>"},
    "k_gemm128": {"k_gemm128<0>", "k_gemm128<1>"},
    "k_diag_inv_b": {"k_diag_inv_b<0>"},
}


def test_every_kernel_in_the_library_is_reachable_from_a_default_path_or_a_documented_fallback():
    import shutil
    import subprocess

    from pykrige_amd import build

    if not shutil.which("nm"):
        pytest.skip("binutils nm not on PATH")
    build.build_library()
    out = subprocess.run(["nm", "-D", "-C", build.OUT], capture_output=True, text=True, check=True).stdout
    inst = sorted({m.group(1) for m in re.finditer(r"__device_stub__([A-Za-z_0-9]+(?:<[^(]*>)?)\(", out)})
    assert len(inst) >= 200, len(inst)
    fam = {}
    for k in inst:
        fam.setdefault(k.split("<")[0], set()).add(k)
    unknown = sorted(set(fam) - set(REACHABLE))
    assert not unknown, "kernels in libmikrige.so that no default path or documented fallback launches (experiments belong in tools/): %s" % unknown
    assert not sorted(set(REACHABLE) - set(fam)), sorted(set(REACHABLE) - set(fam))
    for name, want in PINNED.items():
        assert fam[name] == want, (name, sorted(fam[name] ^ want))
    # and the device headers of the library stay the size the prune left them (round-5 review: mik_k_inverse.h <= 1200 lines)
    n = len(open(os.path.join(ROOT, "pykrige_amd", "csrc", "mik_k_inverse.h")).read().split("\n"))
    assert n <= 1200, n


def test_every_option_and_environment_variable_of_the_library_is_documented():
    """mik_set_option keys and MIK_* environment variables the library reads (csrc/mikrige.hip) appear in the header's option
    list (include/mikrige.h) / INTEGRATION.md's environment table: the boundary's documentation cannot fall behind the code."""
    csrc = os.path.join(ROOT, "pykrige_amd", "csrc")
    src = "".join(open(os.path.join(csrc, f)).read() for f in sorted(os.listdir(csrc)) if f.endswith(".hip"))  # every translation unit (round 5)
    header = open(os.path.join(ROOT, "include", "mikrige.h")).read()
    integration = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    keys = sorted(set(re.findall(r'strcmp\(key, "([a-z_0-9]+)"\)', src)))
    assert len(keys) >= 18
    for k in keys:
        assert '"%s"' % k in header, "option %r is not documented in include/mikrige.h" % k
    envs = sorted(set(re.findall(r'getenv\("(MIK_[A-Z_0-9]+)"\)', src)))
    assert "MIK_NGPU" in envs and "MIK_PANEL_STREAM" in envs
    for e in envs:
        assert "`%s`" % e in integration or "`%s`" % e in header or e in integration, "%s is not in INTEGRATION.md's environment table" % e


def test_ctypes_structs_match_header_layout():
    from pykrige_amd import _lib

    # mik_problem: 2 x int32, int64, 4 pointers, 3 doubles, double, 4 x int32, 3 pointers = 120 bytes on LP64
    assert ctypes.sizeof(_lib.MikProblem) == 4 + 4 + 8 + 4 * 8 + 3 * 8 + 8 + 4 * 4 + 3 * 8 + 2 * 4
    assert ctypes.sizeof(_lib.MikPoints) == 8 + 5 * 8
    assert ctypes.sizeof(_lib.MikTiming) == 5 * 8 + 8 + 8 + 4 * 4 + 8 + 2 * 4 + 8 + 2 * 4 + 6 * 4 + 3 * 8 + 2 * 4 + 5 * 8 + 8 + 2 * 4 + 8 + 2 * 4 + 8  # (ABI 7: + exchange_bytes)
    # mik_grid: 2 x int32, 3 x int64, 3 pointers, 3 + 9 + 3 doubles, 2 x int64, 2 pointers
    assert ctypes.sizeof(_lib.MikGrid) == 2 * 4 + 3 * 8 + 3 * 8 + 15 * 8 + 2 * 8 + 2 * 8


def test_no_gpu_means_a_loud_error_not_a_fallback():
    import pykrige_amd as pa
    from pykrige_amd import _lib

    if _lib.load().mik_device_count() > 0:
        pytest.skip("a GPU is visible")
    ok = pa.OrdinaryKriging([0.0, 1.0, 2.0], [0.0, 1.0, 0.5], [1.0, 2.0, 3.0], variogram_model="linear",
                            variogram_parameters=[1.0, 0.1])
    with pytest.raises(RuntimeError, match="no HIP device"):
        ok.execute("grid", [0.0, 1.0], [0.0, 1.0])


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "pykrige_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", text, flags=re.M), f
                assert "kriging_oracle" not in text, f


def test_parameter_list_rules():
    from pykrige_amd import core

    assert core.make_variogram_parameter_list("spherical", [1.0, 0.5, 0.1]) == [0.9, 0.5, 0.1]  # list form = FULL sill
    assert core.make_variogram_parameter_list("gaussian", {"psill": 2.0, "range": 3.0, "nugget": 0.5}) == [2.0, 3.0, 0.5]
    assert core.make_variogram_parameter_list("exponential", {"sill": 2.0, "range": 3.0, "nugget": 0.5}) == [1.5, 3.0, 0.5]
    assert core.make_variogram_parameter_list("linear", {"slope": 2.0, "nugget": 0.5}) == [2.0, 0.5]
    assert core.make_variogram_parameter_list("power", [1.0, 1.5, 0.0]) == [1.0, 1.5, 0.0]
    assert core.make_variogram_parameter_list("linear", None) is None
    with pytest.raises(ValueError):
        core.make_variogram_parameter_list("linear", [1.0, 2.0, 3.0])
    with pytest.raises(KeyError):
        core.make_variogram_parameter_list("spherical", {"range": 1.0, "nugget": 0.0})
    with pytest.raises(TypeError):
        core.make_variogram_parameter_list("spherical", (1.0, 2.0, 3.0))


def test_host_anisotropy_is_bit_identical_to_the_oracle_restatement():
    from pykrige_amd import core

    rng = np.random.default_rng(0)
    X2, X3 = rng.random((500, 2)) * 7 - 3, rng.random((500, 3)) * 7 - 3
    a = core.adjust_for_anisotropy(X2, [0.3, -0.2], [3.0], [45.0])
    b = ko.adjust_for_anisotropy(X2, [0.3, -0.2], [3.0], [45.0])
    assert np.array_equal(a, b)
    a = core.adjust_for_anisotropy(X3, [0.3, -0.2, 1.0], [1.5, 2.0], [10.0, 20.0, 30.0])
    b = ko.adjust_for_anisotropy(X3, [0.3, -0.2, 1.0], [1.5, 2.0], [10.0, 20.0, 30.0])
    assert np.array_equal(a, b)
    # known answers of tests/test_core.py:83-111
    x = np.array([1.0, 0.0, -1.0, 0.0])
    y = np.array([0.0, 1.0, 0.0, -1.0])
    r = core.adjust_for_anisotropy(np.vstack((x, y)).T, [0.0, 0.0], [2.0], [90.0])
    np.testing.assert_allclose(r[:, 0], [0.0, 1.0, 0.0, -1.0], atol=1e-12)
    np.testing.assert_allclose(r[:, 1], [-2.0, 0.0, 2.0, 0.0], atol=1e-12)


def test_bilinear_external_z():
    from pykrige_amd import core

    gx, gy = np.array([0.0, 1.0, 2.0, 4.0]), np.array([10.0, 20.0, 40.0])
    zg = np.add.outer(gy * 0.5, gx * 2.0)  # plane: bilinear interpolation is exact
    x = np.array([0.0, 0.5, 2.0, 3.3, 4.0, 1.0])
    y = np.array([10.0, 12.0, 40.0, 25.0, 20.0, 20.0])
    np.testing.assert_allclose(core.bilinear_zscalars(zg, gx, gy, x, y), y * 0.5 + x * 2.0, rtol=1e-13)
    with pytest.raises(ValueError):
        core.bilinear_zscalars(zg, gx, gy, np.array([5.0]), np.array([10.0]))


def test_constructor_and_execute_argument_validation():
    import pykrige_amd as pa

    x, y, v = [0.0, 1.0, 2.0, 0.5], [0.0, 1.0, 0.5, 2.0], [1.0, 2.0, 3.0, 0.0]
    with pytest.raises(ValueError):
        pa.OrdinaryKriging(x, y, v, variogram_model="blurg")
    with pytest.raises(ValueError):
        pa.OrdinaryKriging(x, y, v, variogram_model="linear", variogram_parameters=[1.0, 0.0], exact_values="blurg")
    with pytest.raises(ValueError):
        pa.OrdinaryKriging(x, y, v, variogram_model="linear", variogram_parameters=[1.0, 0.0], pseudo_inv_type="qr")
    with pytest.raises(ValueError):
        pa.UniversalKriging(x, y, v, variogram_model="linear", variogram_parameters=[1.0, 0.0], drift_terms=["point_log"])
    with pytest.raises(TypeError):
        pa.UniversalKriging(x, y, v, variogram_model="linear", variogram_parameters=[1.0, 0.0], drift_terms=["specified"],
                            specified_drift=np.zeros(4))
    ok = pa.OrdinaryKriging(x, y, v, variogram_model="linear", variogram_parameters=[1.0, 0.0])
    for bad in (dict(style="blurg"), dict(style="grid", backend="mystery")):
        with pytest.raises(ValueError):
            ok.execute(bad.get("style", "grid"), [0.0, 1.0], [0.0, 1.0], backend=bad.get("backend", "vectorized"))
    with pytest.raises(IOError):
        ok.execute("masked", [0.0, 1.0], [0.0, 1.0])
    with pytest.raises(ValueError):
        ok.execute("points", [0.0, 1.0], [0.0])
    with pytest.raises(ValueError):
        ok.execute("grid", [0.0], [0.0], n_closest_points=1)
    fit = pa.OrdinaryKriging(np.random.default_rng(1).random(40), np.random.default_rng(2).random(40),
                             np.random.default_rng(3).random(40), variogram_model="spherical")
    assert len(fit.variogram_model_parameters) == 3 and all(np.isfinite(fit.variogram_model_parameters))


def test_variogram_fit_matches_reference_fits():
    """Constructor-time fit (host, not the hot path) against parameters the real reference fitted
    (tests/golden/fit_variograms.npz, oracle/make_golden_extra.py)."""
    import pykrige_amd as pa
    from tests import _fixtures as fx

    g = fx.load("fit_variograms")
    for model in ("linear", "power", "gaussian", "spherical", "exponential", "hole-effect"):
        for weight in (False, True):
            key = "%s_%d" % (model.replace("-", ""), int(weight))
            ok = pa.OrdinaryKriging(g["x"], g["y"], g["v"], variogram_model=model, nlags=8, weight=weight,
                                    anisotropy_scaling=2.0, anisotropy_angle=30.0)
            np.testing.assert_allclose(ok.lags, g["lags_" + key], rtol=1e-12)
            np.testing.assert_allclose(ok.semivariance, g["semi_" + key], rtol=1e-12)
            np.testing.assert_allclose(ok.variogram_model_parameters, g["par_" + key], rtol=1e-6, atol=1e-9)
    ok3 = pa.OrdinaryKriging3D(g["x3"], g["y3"], g["z3"], g["v3"], variogram_model="spherical", nlags=6)
    np.testing.assert_allclose(ok3.variogram_model_parameters, g["par_3d"], rtol=1e-6, atol=1e-9)


def test_geographic_variogram_fit_matches_reference():
    import pykrige_amd as pa
    from tests import _fixtures as fx

    g = fx.load("geo_ok2d")
    ok = pa.OrdinaryKriging(g["x"], g["y"], g["v"], variogram_model="spherical", coordinates_type="geographic", nlags=7)
    np.testing.assert_allclose(ok.lags, g["fit_lags"], rtol=1e-12)
    np.testing.assert_allclose(ok.semivariance, g["fit_semi"], rtol=1e-12)
    np.testing.assert_allclose(ok.variogram_model_parameters, g["fit_par"], rtol=1e-6, atol=1e-9)
    with pytest.raises(ValueError):
        pa.OrdinaryKriging(g["x"], g["y"], g["v"], variogram_model="linear", variogram_parameters=[1.0, 0.0], coordinates_type="martian")


def test_sklearn_side_host_pieces_without_a_gpu():
    """compat / rk / ck host logic that needs no device: method validation, keyword routing per method (compat.py:37-74,
    196-229), the ilr transformation pair (ck.py:215-291), learner type checks (compat.py:294-307)."""
    pytest.importorskip("sklearn")
    from sklearn.linear_model import LinearRegression, LogisticRegression

    from pykrige_amd import ck, compat, rk

    with pytest.raises(ValueError):
        compat.Krige(method="nope")
    k = compat.Krige(method="universal3d", anisotropy_scaling=(2.0, 3.0), anisotropy_angle=(10.0, 20.0, 30.0), drift_terms=["regional_linear"])
    kw = k._method_specific()
    assert kw == dict(anisotropy_scaling_y=2.0, anisotropy_scaling_z=3.0, anisotropy_angle_x=10.0, anisotropy_angle_y=20.0,
                      anisotropy_angle_z=30.0, drift_terms=["regional_linear"], functional_drift=None)
    assert set(compat.Krige(method="ordinary")._method_specific()) == {"anisotropy_scaling", "anisotropy_angle", "enable_statistics", "coordinates_type"}
    assert set(k.get_params()) >= {"method", "n_closest_points", "ext_drift_grid", "pseudo_inv_type"}  # sklearn cloning works
    with pytest.raises(ValueError):
        k._dimensionality_check(np.zeros((4, 2)))
    assert list(compat.Krige()._dimensionality_check(np.ones((4, 2)), ext="points")) == ["xpoints", "ypoints"]
    with pytest.raises(Exception, match="Not trained"):
        compat.Krige().predict(np.zeros((2, 2)))
    rng = np.random.default_rng(0)
    comp = ck.closure(rng.random((20, 4)) + 0.05)
    assert np.allclose(comp.sum(1), 1.0)
    coords = ck.ilr_transformation(comp)
    assert coords.shape == (20, 3) and np.allclose(ck.inverse_ilr_transformation(coords), comp, atol=1e-13)
    onehot = np.eye(3)[[0, 2, 1]]
    assert np.all(np.isfinite(ck.ilr_transformation(onehot)))  # zeros are lifted to eps, not -inf
    compat.check_sklearn_model(LinearRegression())
    compat.check_sklearn_model(LogisticRegression(), task="classification")
    with pytest.raises(RuntimeError):
        compat.check_sklearn_model(LinearRegression(), task="classification")
    with pytest.raises(RuntimeError):
        rk.RegressionKriging(regression_model=object())
    with pytest.raises(ValueError):
        ck.ClassificationKriging(classification_model=LogisticRegression(), method="bad")


def test_kriging_tools_grid_files_match_the_reference(tmp_path):
    """kriging_tools (reference kriging_tools.py:23-459): files written here are byte-identical to the ones the real
    reference wrote (tests/golden/tools/, oracle/make_golden_extra.py --tools; ZMAP date / file-name comment lines aside),
    and both readers return what the reference's readers returned, header variants and footer skipping included."""
    from pykrige_amd import kriging_tools as kt

    gdir = os.path.join(ROOT, "tests", "golden", "tools")
    g = np.load(os.path.join(ROOT, "tests", "golden", "tools_grid_files.npz"))
    x, y, xs, ys, z, mask, zz = (g[k] for k in ("x", "y", "xs", "ys", "z", "mask", "zz"))
    kt.write_asc_grid(x, y, z, str(tmp_path / "a.asc"), style=1)
    assert open(tmp_path / "a.asc").read() == open(os.path.join(gdir, "style1.asc")).read()
    kt.write_asc_grid(xs, ys, np.ma.array(z, mask=mask), str(tmp_path / "b.asc"), no_data=-9999.0, style=2)
    assert open(tmp_path / "b.asc").read() == open(os.path.join(gdir, "style2_masked.asc")).read()
    kt.write_zmap_grid(x, y, np.ma.array(zz, mask=mask), str(tmp_path / "masked.zmap"), coord_sys="EPSG:1234")

    def stable(path):
        return [line for line in open(path) if "CREATION" not in line]

    assert stable(tmp_path / "masked.zmap") == stable(os.path.join(gdir, "masked.zmap"))
    for tag, args in (("style1", ("style1.asc",)), ("style2", ("style2_masked.asc",)), ("variant", ("variant_header.asc", 2))):
        grid, gx, gy, cell, nod = kt.read_asc_grid(os.path.join(gdir, args[0]), *args[1:])
        assert np.array_equal(grid, g[tag + "_grid"]) and np.array_equal(gx, g[tag + "_x"]) and np.array_equal(gy, g[tag + "_y"])
        assert np.array_equal(np.atleast_1d(np.asarray(cell, dtype=float)), g[tag + "_cell"]) and nod == float(g[tag + "_nodata"])
    grid, gx, gy, cell, nod, cs = kt.read_zmap_grid(os.path.join(gdir, "masked.zmap"))
    assert np.array_equal(grid, g["zmap_grid"]) and np.array_equal(gx, g["zmap_x"]) and np.array_equal(gy, g["zmap_y"])
    assert np.array_equal(np.asarray(cell), g["zmap_cell"]) and nod == float(g["zmap_nodata"]) and cs == str(g["zmap_cs"])
    with pytest.raises(ValueError):  # irregular spacing, bad style, non-square cells for style 2
        kt.write_asc_grid([0.0, 1.0, 3.0], y, np.zeros((7, 3)), str(tmp_path / "c.asc"))
    with pytest.raises(ValueError):
        kt.write_asc_grid(x, y, z, str(tmp_path / "c.asc"), style=3)
    with pytest.raises(ValueError):
        kt.write_asc_grid(x, y, z, str(tmp_path / "c.asc"), style=2)
    with pytest.raises(IOError):
        open(tmp_path / "bad.asc", "w").write("ncols 2\nbogus 3\n")
        kt.read_asc_grid(str(tmp_path / "bad.asc"))


def test_plot_helpers_and_enable_plotting_without_a_gpu(monkeypatch):
    """display_variogram_model / plot_epsilon_residuals / enable_plotting (ok.py:355-356, 555-567, 601-609): host-side
    matplotlib calls; the constructor shows the variogram when enable_plotting is set."""
    matplotlib = pytest.importorskip("matplotlib")
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt

    import pykrige_amd as pa

    shown = []
    monkeypatch.setattr(plt, "show", lambda *a, **k: shown.append(len(plt.gcf().axes[0].lines) + len(plt.gcf().axes[0].collections)))
    rng = np.random.default_rng(1)
    x, y, v = rng.random(40), rng.random(40), rng.random(40)
    ok = pa.OrdinaryKriging(x, y, v, variogram_model="spherical", enable_plotting=True)
    assert shown == [2]  # binned points + model curve, once, from the constructor
    ok.switch_plotting()
    ok.update_variogram_model("exponential")
    assert shown == [2]
    ok.display_variogram_model()
    assert shown == [2, 2]
    ok.__dict__["epsilon"] = rng.standard_normal(39)  # the statistics themselves need the device
    ok.plot_epsilon_residuals()
    assert shown == [2, 2, 2]  # scatter + zero line
    plt.close("all")


def test_module_paths_of_the_reference_resolve():
    """`from pykrige.ok import OrdinaryKriging` etc. keep working with the package name swapped (pykrige/__init__.py:41-45)."""
    import pykrige_amd
    from pykrige_amd.ok import OrdinaryKriging
    from pykrige_amd.ok3d import OrdinaryKriging3D
    from pykrige_amd.uk import UniversalKriging
    from pykrige_amd.uk3d import UniversalKriging3D

    assert OrdinaryKriging is pykrige_amd.OrdinaryKriging and UniversalKriging is pykrige_amd.UniversalKriging
    assert OrdinaryKriging3D is pykrige_amd.OrdinaryKriging3D and UniversalKriging3D is pykrige_amd.UniversalKriging3D
    assert pykrige_amd.kt.write_asc_grid is pykrige_amd.kriging_tools.write_asc_grid
    import pykrige_amd.compat, pykrige_amd.core, pykrige_amd.variogram_models  # noqa: F401


def test_bare_import_binds_the_submodules_like_upstream():
    """`import pykrige; pykrige.ok.OrdinaryKriging(...)` works upstream (pykrige/__init__.py:44-48 imports from the submodules, which binds them,
    and lists them in __all__): the same with the package name swapped, in a fresh interpreter."""
    import subprocess
    import sys

    code = ("import pykrige_amd as p; assert p.ok.OrdinaryKriging is p.OrdinaryKriging and p.uk.UniversalKriging is p.UniversalKriging; "
            "assert p.ok3d.OrdinaryKriging3D is p.OrdinaryKriging3D and p.uk3d.UniversalKriging3D is p.UniversalKriging3D; "
            "assert p.kriging_tools is p.kt and p.__version__; assert all(hasattr(p, n) for n in p.__all__)")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-1500:]


def test_narrow_float_points_are_centred_as_the_reference_centres_them():
    """The reference never casts the prediction coordinates (ok.py:849-850) and centres them in place (core.py:146 `X -= center`): float32 coordinate
    arrays are rounded to float32 AFTER the subtraction (float64 arithmetic, float32 store), then rotated in float64.  _narrow_dtype / _as_centred restate
    that for every route the points take (host adjustment, raw columns and grid axes for the device); the differential run against the real reference is
    tests/test_input_forms_vs_reference.py (GPU).  Also here: 2-D point arrays, which upstream's cdist refuses."""
    import pykrige_amd as pa
    from pykrige_amd import core

    rng = np.random.default_rng(5)
    x, y, v = rng.random(30) * 7 + 3, rng.random(30) * 2 - 9, rng.random(30)
    m = pa.OrdinaryKriging(x, y, v, variogram_model="exponential", variogram_parameters=[1.0, 2.0, 0.1], anisotropy_scaling=2.5, anisotropy_angle=33.0)
    px, py = (rng.random(200) * 7 + 3).astype(np.float32), (rng.random(200) * 2 - 9).astype(np.float32)
    # the reference's operations, as written (core.py:142-146, 190-193)
    X = np.vstack((px, py)).T
    assert X.dtype == np.float32
    X -= np.asarray(m._center())[None, :]
    rot, stretch = core.anisotropy_matrices(2, m._scaling(), m._angle())
    want = np.dot(np.diag(stretch), np.dot(rot, X.T)).T + np.asarray(m._center())[None, :]  # (anisotropy_matrices returns the stretch diagonal)
    assert m._narrow_dtype((px, py)) == np.float32
    assert m._narrow_dtype((px, py.astype(np.float64))) is None and m._narrow_dtype((px.tolist(), py.tolist())) is None  # (Python floats are float64)
    assert m._narrow_dtype((list(px), list(py))) == np.float32  # (a list of np.float32 scalars becomes a float32 array upstream as well)
    assert m._narrow_dtype((np.arange(3), np.arange(3))) is None and m._narrow_dtype((0.5, 0.25)) is None
    for style_axes, route in (((px, py), "points"),):
        P = m._prepare(route, style_axes, None)
        got_raw = np.stack([np.asarray(P.arrays[:, 0]), np.asarray(P.arrays[:, 1])], axis=1)  # raw columns: what the device adjusts
        got = core.adjust_for_anisotropy(got_raw.copy(), m._center(), m._scaling(), m._angle())
        plain = core.adjust_for_anisotropy(np.stack([px.astype(np.float64), py.astype(np.float64)], axis=1), m._center(), m._scaling(), m._angle())
        assert np.abs(got - want).max() <= 4e-15, np.abs(got - want).max()
        assert np.abs(plain - want).max() > 1e-8  # (what kriging the float32 values as exact doubles would have been off by)
    os.environ["MIK_DEVICE_POINTS"] = "0"  # the host-adjustment route
    try:
        P = m._prepare("points", (px, py), None)
        assert np.abs(np.asarray(P.arrays) - want).max() <= 4e-15
    finally:
        del os.environ["MIK_DEVICE_POINTS"]
    gx, gy = np.linspace(3, 10, 9, dtype=np.float32), np.linspace(-9, -7, 5, dtype=np.float32)
    P = m._prepare("grid", (gx, gy), None)  # grid axes for the device: centring is per axis
    for got, ax, c in zip(P.axes, (gx, gy), m._center()):
        assert np.array_equal(got, (ax.astype(np.float64) - c).astype(np.float32).astype(np.float64) + c)
    assert m._prepare("grid", (gx.astype(np.float64), gy), None).axes[1].tolist() == gy.astype(np.float64).tolist()  # mixed dtypes promote to float64: no rounding
    geo = pa.OrdinaryKriging(x, y, v, variogram_model="exponential", variogram_parameters=[1.0, 2.0, 0.1], coordinates_type="geographic")
    assert geo._narrow_dtype((px, py)) is None  # geographic coordinates are not centred (ok.py:892-896)
    with pytest.raises(ValueError):
        m._prepare("points", (px.reshape(20, 10), py.reshape(20, 10)), None)  # cdist refuses them upstream
    # (a single station: upstream cannot construct the object -- np.amax over its empty pair distances, core.py:465, an accident like the integer
    #  axes -- ; here it kriges: weight 1 everywhere, test_one_shot_c_entry_point_and_degenerate_inputs)
    pa.OrdinaryKriging(x[:1], y[:1], v[:1], variogram_model="exponential", variogram_parameters=[1.0, 2.0, 0.1])


def test_anisotropy_adjustment_is_bit_identical_to_the_reference():
    """core.adjust_for_anisotropy (reference core.py:120-193): same NumPy operations in the same order, so the adjusted
    coordinates -- which decide the `abs(bd) <= eps` exact-hit rule -- equal the real reference's bit for bit
    (tests/golden/aniso_adjust.npz); the oracle's own restatement is held to the same."""
    from oracle import kriging_oracle as ko
    from pykrige_amd import core

    g = np.load(os.path.join(ROOT, "tests", "golden", "aniso_adjust.npz"))
    for d in ("2", "3"):
        X, c, s, a, Y = g["X" + d], list(g["c" + d]), list(g["s" + d]), list(g["a" + d]), g["Y" + d]
        assert np.array_equal(core.adjust_for_anisotropy(X.copy(), c, s, a), Y)
        assert np.array_equal(ko.adjust_for_anisotropy(X.copy(), c, s, a), Y)


def test_point_lists_go_to_the_device_raw_unless_the_host_needs_them_adjusted(monkeypatch):
    """Host routing of execute()'s point branch (no GPU): coordinate arrays are handed over RAW with the anisotropy matrices
    (mik_adjust_points does what ok.py:879-885 does on the host) -- except with functional drifts (their callables take the adjusted
    coordinates, uk.py:1290-1296), geographic coordinates (never adjusted, ok.py:892-896) and MIK_DEVICE_POINTS=0; the matrices
    reproduce core.adjust_for_anisotropy entry for entry."""
    import pykrige_amd as pa
    from pykrige_amd import core

    rng = np.random.default_rng(3)
    n = 60
    x, y, zc = rng.random(n) * 10, rng.random(n) * 6, rng.random(n) * 3
    v = np.sin(x) + 0.3 * y
    px, py, pz = rng.random(40) * 10, rng.random(40) * 6, rng.random(40) * 3
    monkeypatch.delenv("MIK_DEVICE_POINTS", raising=False)
    monkeypatch.delenv("MIK_DEVICE_GRID", raising=False)
    ok = pa.OrdinaryKriging(x, y, v, variogram_model="exponential", variogram_parameters=[1.0, 4.0, 0.01], anisotropy_scaling=2.5,
                            anisotropy_angle=33.0)
    want = core.adjust_for_anisotropy(np.stack((px, py), 1), ok._center(), ok._scaling(), ok._angle())
    P = ok._prepare("points", (px, py), None)
    assert P.raw and np.array_equal(P.arrays, np.stack((px, py), 1))
    Y = np.dot(np.diag(P.stretch), np.dot(P.rot, (P.arrays - np.asarray(P.center)[None, :]).T)).T + np.asarray(P.center)[None, :]
    assert np.array_equal(Y, want)
    monkeypatch.setenv("MIK_DEVICE_POINTS", "0")
    P = ok._prepare("points", (px, py), None)
    assert not P.raw and np.array_equal(P.arrays, want)
    monkeypatch.delenv("MIK_DEVICE_POINTS")
    uk = pa.UniversalKriging(x, y, v, variogram_model="spherical", variogram_parameters=[1.0, 5.0, 0.0], anisotropy_scaling=0.6,
                             anisotropy_angle=-71.0, drift_terms=["regional_linear", "specified"], specified_drift=[x * y])
    P = uk._prepare("points", (px, py), None, [px * py], "vectorized")
    assert P.raw and P.extra.shape == (1, 40) and np.array_equal(P.extra[0], px * py)
    uf = pa.UniversalKriging(x, y, v, variogram_model="spherical", variogram_parameters=[1.0, 5.0, 0.0], anisotropy_scaling=0.6,
                             anisotropy_angle=-71.0, drift_terms=["functional"], functional_drift=[lambda a, b: a * b])
    P = uf._prepare("points", (px, py), None, None, "vectorized")
    adj = core.adjust_for_anisotropy(np.stack((px, py), 1), uf._center(), uf._scaling(), uf._angle())
    assert not P.raw and np.array_equal(P.arrays, adj) and np.array_equal(P.extra[0], adj[:, 0] * adj[:, 1])
    k3 = pa.OrdinaryKriging3D(x, y, zc, v, variogram_model="gaussian", variogram_parameters=[1.0, 3.0, 0.05], anisotropy_scaling_y=1.5,
                              anisotropy_scaling_z=0.7, anisotropy_angle_x=10.0, anisotropy_angle_y=20.0, anisotropy_angle_z=30.0)
    P = k3._prepare("points", (px, py, pz), None)
    assert P.raw and P.arrays.shape == (40, 3) and np.asarray(P.rot).shape == (3, 3) and list(P.stretch) == [1.0, 1.5, 0.7]
    geo = pa.OrdinaryKriging(x * 10, y * 10, v, variogram_model="linear", variogram_parameters=[1.0, 0.0], coordinates_type="geographic")
    assert not geo._prepare("points", (px, py), None).raw
    # a grid built on the host (MIK_DEVICE_GRID=0) is a point list like any other
    monkeypatch.setenv("MIK_DEVICE_GRID", "0")
    gx, gy = np.linspace(0, 10, 7), np.linspace(0, 6, 5)
    mask = rng.random((5, 7)) < 0.3
    P = ok._prepare("masked", (gx, gy), mask)
    assert P.raw and P.arrays.shape == (35, 2) and P.mask.shape == (35,)


def _kernel_metadata():
    """{mangled kernel name: {vgpr_count, agpr_count, private_segment_fixed_size, group_segment_fixed_size}} of the gfx950 code object
    inside the built library (the clang offload bundle in .hip_fatbin, read with llvm-readelf --notes)."""
    import struct
    import subprocess
    import tempfile

    readelf = "/opt/rocm/lib/llvm/bin/llvm-readelf"
    lib = os.path.join(ROOT, "pykrige_amd", "libmikrige.so")
    if not (os.path.exists(readelf) and os.path.exists(lib)):
        pytest.skip("needs the built library and ROCm's llvm-readelf")
    blob = open(lib, "rb").read()
    elfs, at = [], blob.find(b"__CLANG_OFFLOAD_BUNDLE__")
    while at >= 0:  # one bundle per translation unit of the library (pykrige_amd/build.py)
        n = struct.unpack_from("<Q", blob, at + 24)[0]
        off = at + 32
        for _ in range(n):
            o, s, ts = struct.unpack_from("<QQQ", blob, off)
            off += 24
            triple = blob[off:off + ts].decode()
            off += ts
            if "gfx950" in triple:
                elfs.append(blob[at + o:at + o + s])
        at = blob.find(b"__CLANG_OFFLOAD_BUNDLE__", at + 24)
    # --offload-compress (round 5, pykrige_amd/build.py): the bundles are zstd-compressed ("CCOB", version 3: magic, u16 version, u16
    # method, u64 file size, u64 uncompressed size, u64 hash); clang-offload-bundler unpacks them
    bundler = "/opt/rocm/lib/llvm/bin/clang-offload-bundler"
    at = blob.find(b"CCOB")
    while at >= 0:
        ver = struct.unpack_from("<H", blob, at + 4)[0]
        assert ver == 3 and os.path.exists(bundler), ("compressed offload bundle version", ver)
        size = struct.unpack_from("<Q", blob, at + 8)[0]
        with tempfile.TemporaryDirectory() as d:
            src, dst = os.path.join(d, "b.hipfb"), os.path.join(d, "b.co")
            open(src, "wb").write(blob[at:at + size])
            subprocess.run([bundler, "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + src, "--output=" + dst],
                           check=True, capture_output=True)
            elfs.append(open(dst, "rb").read())
        at = blob.find(b"CCOB", at + size)
    assert elfs and all(e[:4] == b"\x7fELF" for e in elfs), "no gfx950 code object in the library"
    notes = ""
    for elf in elfs:
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(elf)
            f.flush()
            notes += subprocess.run([readelf, "--notes", f.name], capture_output=True, text=True, check=True).stdout
    out = {}
    for block in notes.split("\n  - .agpr_count:")[1:]:
        block = ".agpr_count:" + block
        get = lambda key: int(re.search(r"\.%s:\s+(\d+)" % key, block).group(1))
        name = re.search(r"\.name:\s+(\S+)", block).group(1)
        out[name] = {k: get(k) for k in ("vgpr_count", "agpr_count", "private_segment_fixed_size", "group_segment_fixed_size")}
    return out


def test_kernel_register_budgets_of_the_built_library():
    """The occupancy the hot kernels were tuned for is a property of the BUILD (registers per wavefront, spills, LDS per block), so it
    is pinned here without a GPU: the contraction and the 8-wave trailing update at 128 VGPRs (4 wavefronts per SIMD = two 8-wave
    blocks per CU, 64 KB of LDS each), no spills inside them beyond the few bytes of per-tile state; every moving-window class the
    dispatcher uses within the register budget of the wavefronts per SIMD its launch bound promises (round 3: the class table's
    cliffs were occupancy cliffs -- 256 VGPRs + AGPRs meant one wavefront per SIMD), none of them parking values in AGPRs."""
    md = _kernel_metadata()

    def one(pattern):
        hits = [v for k, v in md.items() if re.fullmatch(pattern, k)]
        assert len(hits) == 1, (pattern, len(hits))
        return hits[0]

    k = one(r"_ZN3mik10k_contractILb1ELi2ELb1ELb0ELb1ELb0EEEv.*")  # symmetric, 8 waves, persistent, triangular diagonal blocks: the default
    assert k["vgpr_count"] <= 128 and k["agpr_count"] == 0 and k["private_segment_fixed_size"] <= 64 and k["group_segment_fixed_size"] <= 65544, k
    # the range-aware contraction over gathered row groups (round 4, second session): the same budget, and no scratch beyond the queue's
    # steal path (a scratch access inside its K loops makes hipcc wait for vmcnt(0) right behind the step's DMA: no overlap at all)
    # {EPI, H8, PROF}: pairs of 8-station tiles with the epilogue from the B tile in LDS (the default) and its profiling instantiation
    # (MIK_SPG_PROF=1); the 16-station and global-memory-epilogue forms left the library in round 6
    for epi, h8, prof in ((1, 1, 0), (1, 1, 1)):
        k = one(r"_ZN3mik14k_contract_spgILi2ELb%dELb%dELb%dEEEvNS_7SpgArgsE" % (epi, h8, prof))
        assert k["vgpr_count"] <= 128 and k["agpr_count"] == 0 and k["private_segment_fixed_size"] <= 32 and k["group_segment_fixed_size"] <= 70100, k  # staging 64 KB + the four wave-rows' sums 4 KB + records
    for sym in (0, 1):
        k = one(r"_ZN3mik8k_updateILb%dELi2EEEv.*" % sym)
        assert k["vgpr_count"] <= 128 and k["agpr_count"] == 0 and k["private_segment_fixed_size"] == 0, k
    # VGPR budget per wavefront for w wavefronts per SIMD (512 registers per lane and SIMD, allocated in blocks of 8)
    budget = {1: 512, 2: 256, 3: 168, 4: 128, 5: 96, 6: 80, 7: 72, 8: 64}
    classes = {(4, 4): 7, (4, 6): 5, (4, 8): 4, (4, 10): 3, (4, 13): 2, (8, 6): 5, (8, 8): 4, (8, 10): 3, (8, 11): 2, (8, 12): 2, (8, 13): 2,
               (16, 7): 4, (16, 8): 4, (16, 9): 3, (16, 10): 3, (16, 11): 2, (16, 12): 2, (16, 13): 2, (16, 14): 2, (32, 8): 4}
    sizes = {}
    for (g, ri), waves in classes.items():
        for model in ("n1", "0", "2", "3", "4"):  # the dynamic form and the four models instantiated as compile-time constants (round 4)
            k = one(r"_ZN3mik9k_mw_cholILi%dELi%dELi%sEEEvNS_6MwArgsE" % (g, ri, model))
            assert k["agpr_count"] == 0 and k["vgpr_count"] <= budget[waves], ((g, ri, model), waves, k)
            assert k["private_segment_fixed_size"] <= 260, ((g, ri, model), k)  # {16,14} spills 33 registers, the others at most a handful


def test_point_lists_are_not_copied_on_the_host():
    """style='points' with coordinates that go to the device raw: _prepare hands the library views of the caller's arrays (kriging._Cols:
    the (n, d) indexing the module uses, no copy of npt x d doubles); a list or an int array is converted once; the host-adjusted route
    (functional drift) still builds its own array."""
    import pykrige_amd as pa
    from pykrige_amd import kriging

    rng = np.random.default_rng(2)
    x, y, v = rng.random(30), rng.random(30), rng.random(30)
    px, py = rng.random(1000), rng.random(1000)
    ok = pa.OrdinaryKriging(x, y, v, variogram_model="linear", variogram_parameters=[1.0, 0.1])
    P = ok._prepare("points", (px, py), None)
    c = P.arrays
    assert isinstance(c, kriging._Cols) and c.shape == (1000, 2) and P.raw
    assert np.shares_memory(c[:, 0], px) and np.shares_memory(c[:, 1], py)
    assert c[:, 0].flags["C_CONTIGUOUS"]
    sub = c[100:300]
    assert sub.shape == (200, 2) and np.shares_memory(sub[:, 1], py) and np.array_equal(sub[:, 1], py[100:300])
    keep = rng.random(1000) < 0.5
    assert np.array_equal(c[keep, 0], px[keep])
    assert np.array_equal(np.asarray(c), np.stack((px, py), 1))
    P2 = ok._prepare("points", (list(px[:5]), np.arange(5)), None)  # converted, same values
    assert np.array_equal(P2.arrays[:, 0], px[:5]) and P2.arrays[:, 1].dtype == np.float64
    with pytest.raises(ValueError):
        ok._prepare("points", (px, py[:10]), None)
    uk = pa.UniversalKriging(x, y, v, variogram_model="linear", variogram_parameters=[1.0, 0.1], drift_terms=["functional"],
                             functional_drift=[lambda a, b: a + b])
    P3 = uk._prepare("points", (px, py), None)
    assert isinstance(P3.arrays, np.ndarray) and not P3.raw and not np.shares_memory(P3.arrays, px)


def test_reference_private_core_helpers_and_their_known_answers():
    """pykrige_amd.core carries the reference's private host helpers under their own names (core.py:100, 120, 196, 379, 538, 582); the
    reference's tests of them hold known answers (test_core.py:184-376 variogram estimation, 2691-2747 great-circle code with
    geopy-derived distances), restated here."""
    from pykrige_amd import core
    from pykrige_amd import variogram_models as vm

    # test_core.py:304-376: fitted parameters (internal order) of hand-made semivariograms
    lag = np.array([1.0, 2.0, 3.0, 4.0])
    for semis, model, fn, weight, want, tol in (
            ([2.05, 2.95, 4.05, 4.95], "linear", vm.linear_variogram_model, False, [0.98, 1.05], 0.01),
            ([2.05, 2.95, 4.05, 4.95], "linear", vm.linear_variogram_model, True, [0.98, 1.05], 0.01),
            ([1.0, 2.8284271, 5.1961524, 8.0], "power", vm.power_variogram_model, False, [1.0, 1.5, 0.0], 0.001),
            ([1.0, 1.4142, 1.7321, 2.0], "power", vm.power_variogram_model, False, [1.0, 0.5, 0.0], 0.001),
            ([1.2642, 1.7293, 1.9004, 1.9634], "exponential", vm.exponential_variogram_model, False, [2.0, 3.0, 0.0], 0.001),
            ([0.5769, 1.4872, 1.9065, 1.9914], "gaussian", vm.gaussian_variogram_model, False, [2.0, 3.0, 0.0], 0.001),
            ([3.33060952, 3.85063879, 3.96667301, 3.99256374], "exponential", vm.exponential_variogram_model, False, [3.0, 2.0, 1.0], 0.001),
            ([2.60487044, 3.85968813, 3.99694817, 3.99998564], "gaussian", vm.gaussian_variogram_model, False, [3.0, 2.0, 1.0], 0.001)):
        res = core._calculate_variogram_model(lag, np.array(semis), model, fn, weight)
        np.testing.assert_allclose(res, want, tol, tol)
    # test_core.py:184-301: parameter counts, coordinate types, and the binned semivariogram of four points on a line
    xy = np.array([[0.0, 0.0], [1.0, 0.5], [2.0, 2.0], [0.3, 1.7], [1.1, 0.2]])
    zz = np.array([1.0, 2.0, 0.5, 1.5, 0.7])
    for model, params, ctype in (("linear", [0.0], "euclidean"), ("spherical", [0.0], "euclidean"), ("spherical", [0.0, 0.0, 0.0], "tacos")):
        with pytest.raises(ValueError):
            core._initialize_variogram_model(xy, zz, model, params, model, 6, False, ctype)
    with pytest.raises(ValueError):  # geographic coordinates are 2-D only
        core._initialize_variogram_model(np.hstack((xy, xy[:, :1])), zz, "linear", [0.0, 0.0], "linear", 6, False, "geographic")
    with pytest.raises(ValueError):
        core._initialize_variogram_model(xy, zz, "custom", None, None, 6, False, "euclidean")
    x = np.array([1.0 + n / np.sqrt(2) for n in range(4)])
    lags, semi, params = core._initialize_variogram_model(np.vstack((x, x)).T, np.arange(1.0, 5.0), "linear", [0.0, 0.0], "linear", 6, False,
                                                          "euclidean")
    np.testing.assert_allclose(lags, [1.0, 2.0, 3.0])
    np.testing.assert_allclose(semi, [0.5, 2.0, 4.5])
    assert params == [0.0, 0.0]
    line = np.array([1.0, 2.0, 3.0, 4.0])
    lags, semi, _ = core._initialize_variogram_model(np.vstack((line, line, line)).T, line, "linear", [0.0, 0.0], "linear", 3, False, "euclidean")
    np.testing.assert_allclose(lags, np.sqrt(3.0) * np.array([1.0, 2.0, 3.0]))
    np.testing.assert_allclose(semi, [0.5, 2.0, 4.5])
    lags, semi, fitted = core._initialize_variogram_model(xy, zz, "linear", None, vm.linear_variogram_model, 3, False, "euclidean")
    assert len(fitted) == 2 and np.all(np.isfinite(fitted))
    # test_core.py:2691-2747: great-circle distances against geopy's, and the chord <-> arc conversion against them
    lon = np.array([7.0, 7.0, 187.0, 73.231])
    lat = np.array([13.23, 13.2301, -13.23, -79.3])
    d_ref = np.array([[0.0, 1e-4, 180.0, 98.744848317171801], [1e-4, 0.0, 179.9999, 98.744946828324345],
                      [180.0, 179.9999, 0.0, 81.255151682828213], [98.744848317171801, 98.744946828324345, 81.255151682828213, 0.0]])
    d = np.array([[core.great_circle_distance(lon[i], lat[i], lon[j], lat[j]) for j in range(4)] for i in range(4)])
    np.testing.assert_allclose(d, d_ref)
    assert np.all(d >= 0.0) and np.all(d <= 180.0) and np.allclose(d, d.T) and np.allclose(np.diag(d), 0.0)
    glon, glat = np.meshgrid(np.linspace(0, 360.0, 20), np.linspace(-90.0, 90.0, 20))
    rad = np.pi / 180.0
    for i in range(4):
        dx = np.cos(rad * glon) * np.cos(rad * glat) - np.cos(rad * lon[i]) * np.cos(rad * lat[i])
        dy = np.sin(rad * glon) * np.cos(rad * glat) - np.sin(rad * lon[i]) * np.cos(rad * lat[i])
        dz = np.sin(rad * glat) - np.sin(rad * lat[i])
        np.testing.assert_allclose(core.great_circle_distance(lon[i], lat[i], glon, glat),
                                   core.euclid3_to_great_circle(np.sqrt(dx**2 + dy**2 + dz**2)), rtol=1e-5)
    # the aliases are the functions the classes use
    assert core._adjust_for_anisotropy is core.adjust_for_anisotropy and core._make_variogram_parameter_list is core.make_variogram_parameter_list
    r = core._variogram_residuals([1.0, 0.0], lag, lag + 0.5, vm.linear_variogram_model, False)
    np.testing.assert_allclose(r, -0.5)


def test_api_surface_of_the_reference_modules_is_present():
    """Every function, class, public method and class attribute the reference's modules of this path define (parsed from
    /root/reference when it is there; the list below otherwise) exists under the same name in pykrige_amd -- except the three
    private kernels of execute() whose arguments ARE the objects this design never builds (the npt x N distance matrix `bd`):
    _exec_vector / _exec_loop / _exec_loop_moving_window are replaced by the device path behind execute() itself."""
    import importlib

    import pykrige_amd as pa
    from pykrige_amd import core, kriging_tools

    replaced = {"_exec_vector", "_exec_loop", "_exec_loop_moving_window"}
    ref = "/root/reference/src/pykrige/"
    if os.path.isdir(ref):
        import ast

        for mod in ("ok", "uk", "ok3d", "uk3d", "variogram_models", "kriging_tools", "compat", "rk", "ck", "core"):
            ours = importlib.import_module("pykrige_amd." + mod)
            for node in ast.parse(open(ref + mod + ".py").read()).body:
                if isinstance(node, ast.FunctionDef):
                    assert hasattr(ours, node.name), (mod, node.name)
                elif isinstance(node, ast.ClassDef):
                    cls = getattr(ours, node.name)
                    for sub in node.body:
                        names = []
                        if isinstance(sub, ast.FunctionDef):
                            names = [sub.name]
                        elif isinstance(sub, ast.Assign):
                            names = [t.id for t in sub.targets if isinstance(t, ast.Name)]
                        for name in names:
                            assert name in replaced or hasattr(cls, name), (mod, node.name, name)
                elif isinstance(node, ast.Assign):
                    for t in node.targets:
                        if isinstance(t, ast.Name) and not t.id.startswith("__"):
                            assert hasattr(ours, t.id), (mod, t.id)
    # the same facts without the reference tree (the GPU box, a user's machine)
    for cls in (pa.OrdinaryKriging, pa.UniversalKriging, pa.OrdinaryKriging3D, pa.UniversalKriging3D):
        assert sorted(cls.variogram_dict) == ["exponential", "gaussian", "hole-effect", "linear", "power", "spherical"]
        for name in ("execute", "update_variogram_model", "display_variogram_model", "get_variogram_points", "switch_verbose", "switch_plotting",
                     "get_epsilon_residuals", "plot_epsilon_residuals", "get_statistics", "print_statistics", "eps"):
            assert hasattr(cls, name), (cls.__name__, name)
    ok = pa.OrdinaryKriging([0.0, 1.0, 2.0], [0.0, 1.0, 0.5], [1.0, 2.0, 3.0], variogram_model="linear", variogram_parameters=[1.0, 0.1])
    # ok.py:253: for a named model `variogram_function` is that model's function (callers plot the fit with it)
    np.testing.assert_allclose(ok.variogram_function(ok.variogram_model_parameters, np.array([1.0, 2.0])), [1.1, 2.1])
    assert core.eps == 1.0e-10 and sorted(core.P_INV) == ["pinv", "pinvh"]
    np.testing.assert_allclose(core.P_INV["pinvh"](np.diag([2.0, 0.0])), np.diag([0.5, 0.0]))
    assert kriging_tools.space_back_to_front(" 12.5  ") == "   12.5" and kriging_tools.space_back_to_front("abc") == "abc"
    with pytest.raises(ValueError):
        kriging_tools.space_back_to_front("   ")
