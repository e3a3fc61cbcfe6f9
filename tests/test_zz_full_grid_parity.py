"""Round 6: parity over the WHOLE grids BASELINE.json names, in the default `-m gpu` set (the stored reference slabs under
tests/golden/fullsize cover 0.1 - 1.7 % of them).

ONE execute('grid') of the drop-in class over the config's own grid -- config 2: all 10^6 points; config 4: 1024 x 1024; config 3:
200 x 200 x 50; config 5: the 4096 x 64 strip of the 4096 x 4096 grid across the cut between the slabs of GPUs 0 and 1 -- against the
REAL reference (oracle/_ref staged by oracle/build_ref.sh; ok.py:650-683, uk.py:922-1009, ok3d.py:624-657 as written upstream,
backend='vectorized') kriging the same grid slab by slab on the box's host cores (bench.full_grid_parity -> oracle/full_grid.py), every
compared point at north_star's bar: |dz| <= 1e-8, |dsigma^2| <= 1e-6.

The GPU side always kriges the whole grid.  The reference side runs as eight reference processes side by side (its _exec_vector is mostly
single-threaded NumPy; 8 BLAS threads each): the whole config-2 grid in 57 s on the GPU box (one process: 171 - 182 s).  It is still bounded in
wall-clock so that the suite finishes under the driver's limit on any box: BUDGET_S seconds per config (MIK_FULLGRID_BUDGET scales them), cut
further when the suite is already late (SUITE_LIMIT_S).  Slabs are visited in van der Corput order, so a bounded run is spread over the whole grid;
the coverage reached is printed and must be at least MIN_COVERAGE.  `python bench.py --full-parity` is the unbounded run
(profiles/r06_full_grid_parity.txt).  This file sorts last so that its budgets see the time the rest of the suite took."""
import os
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle import ref_package as rp  # noqa: E402

from . import conftest  # noqa: E402

Z_TOL, SS_TOL = 1e-8, 1e-6
BUDGET_S = {2: 150.0, 4: 150.0, 3: 100.0, 5: 150.0}  # whole grids on the GPU box (EPYC 9575F, 256 logical CPUs, 8 reference processes): see the test's output
ORDER = [2, 4, 3, 5]
SUITE_LIMIT_S = 1050.0  # the driver gives `pytest -m gpu` 1200 s
MIN_COVERAGE = 0.02  # a slow box still checks more of every grid than the stored slab did


def _budget(cno):
    scale = float(os.environ.get("MIK_FULLGRID_BUDGET", "1"))
    left = SUITE_LIMIT_S - (time.time() - conftest.SESSION_T0)
    later = sum(BUDGET_S[c] for c in ORDER[ORDER.index(cno):])
    return max(5.0, min(BUDGET_S[cno] * scale, left * BUDGET_S[cno] / later))


@pytest.mark.gpu
@pytest.mark.parametrize("cno", ORDER)
def test_whole_grid_against_the_reference(cno):
    if not rp.available():
        pytest.fail("oracle/_ref is not staged: run oracle/build_ref.sh (or __graft_entry__.build()) where /root/reference exists")
    budget = _budget(cno)
    res = bench.full_grid_parity(cno, budget_s=budget)
    print("\nconfig %d (%s): grid %s, GPU execute %.2f s (%s contraction), cond_1 %.2e; reference checked %d of %d points (%.1f %%, %d of %d slabs, "
          "%.0f s of a %.0f s budget, %.0f points/s with %d reference processes on %d logical CPUs): max|dz| %.2e at %s, max|dss| %.2e at %s%s" % (
              cno, res["workload"], "x".join(map(str, res["grid"])), res["gpu_execute_s"], res["gpu_contraction"], res["cond_1"],
              res["points_checked"], res["points_total"], 100.0 * res["coverage"], res["slabs_checked"], res["slabs_total"], res["reference_s"],
              budget, res["reference_points_per_s"], res["reference_processes"], res["host_cpus"], res["max_abs_dz"], res["worst_dz_at"], res["max_abs_dss"], res["worst_dss_at"],
              "; reference C vs vectorized: %.1e / %.1e" % (res["reference_c_vs_vectorized_max_abs_dz"], res["reference_c_vs_vectorized_max_abs_dss"])
              if "reference_c_vs_vectorized_max_abs_dz" in res else ""))
    assert res["max_abs_dz"] <= Z_TOL and res["max_abs_dss"] <= SS_TOL, res
    assert res["coverage"] >= MIN_COVERAGE, res
    assert res["gpu_contraction"] == ("range-aware" if cno == 5 else "dense")
