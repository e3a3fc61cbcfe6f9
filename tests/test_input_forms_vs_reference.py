"""The drop-in classes and the REAL reference (oracle/_ref, staged by oracle/build_ref.sh) fed the same unusual INPUT FORMS -- float32 / integer / strided /
list / scalar coordinates, empty and one-element point lists, 2-D point arrays, masks of every kind and order, windows at their limits, one to three
stations, duplicated stations with and without pseudo_inv, far-off coordinates, every variogram model, every drift kind on every backend, non-finite coordinates, narrow dtypes into the drifts: either both
return (|dz| <= 1e-8, |dsigma^2| <= 1e-6, same shapes, dtypes, masked-array-ness and masks) or both raise the same exception type.
scripts/edge_forms_vs_reference.py is the list of cases (201 of them); it runs in a process of its own because one form -- a window larger than the
station count on backend='C' -- makes the reference's compiled loop corrupt the heap (left out there, see the script)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_unusual_input_forms_behave_as_in_the_reference():
    sys.path.insert(0, ROOT)
    from oracle import ref_package as rp

    if not (rp.available() and rp.c_available()):
        pytest.skip("the staged reference (oracle/_ref) is not here")
    r = subprocess.run([sys.executable, "-u", os.path.join(ROOT, "scripts", "edge_forms_vs_reference.py")], cwd=ROOT, capture_output=True, text=True, timeout=900)
    lines = [ln for ln in r.stdout.splitlines() if ln and not ln.startswith(" ")]
    agree = [ln for ln in lines if "agree, shape" in ln or "both raise" in ln]
    bad = [ln for ln in lines if "DISAGREE" in ln or "DIFFERENT TYPES" in ln]
    assert r.returncode == 0 and not bad, "\n".join(bad + lines[-3:] + [r.stderr[-2000:]])
    assert len(agree) >= 185, len(agree)  # (the cases that compare or raise alike; the rest are notes: forms the reference itself fails on by accident)


@pytest.mark.gpu
def test_constructed_objects_carry_the_reference_attributes():
    """Every attribute of the reference's instances (name, type, dtype, shape, value: X_ADJUSTED, lags, semivariance, the FITTED variogram parameters as
    the ndarray the least-squares solver returns, delta / sigma / epsilon, Q1 / Q2 / cR ...) and its accessor methods, for eight constructor forms."""
    sys.path.insert(0, ROOT)
    from oracle import ref_package as rp

    if not rp.available():
        pytest.skip("the staged reference (oracle/_ref) is not here")
    r = subprocess.run([sys.executable, "-u", os.path.join(ROOT, "scripts", "attributes_vs_reference.py")], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1500:]
    assert r.stdout.count("missing []; differing []") == 8, r.stdout[-3000:]
