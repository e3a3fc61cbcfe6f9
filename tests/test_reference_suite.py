"""The reference's OWN test-suite judging the drop-in: /root/reference/tests/test_core.py (48 tests), test_api.py and the
scikit-learn wrappers' tests (test_regression_krige.py, test_classification_krige.py) run unmodified
with `pykrige` -> `pykrige_amd` (an import shim in a conftest.py written next to the unpacked tests).  The suite is staged by
oracle/build_ref.sh as oracle/_ref/reference_tests.zip (git-ignored build output; absent -> skipped) and unpacked into pytest's
tmp_path only.  Every test that does not pass is listed below with the reason; nothing else may fail."""
import os
import subprocess
import sys
import xml.etree.ElementTree as ET
import zipfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TESTS_ZIP = os.path.join(ROOT, "oracle", "_ref", "reference_tests.zip")

SHIM = '''
import os, sys
sys.path.insert(0, %r)
import pykrige_amd
from pykrige_amd import ck, compat, core, kriging_tools, ok, ok3d, rk, uk, uk3d, variogram_models  # noqa: F401
sys.modules["pykrige"] = pykrige_amd
for _n in ("ck", "compat", "core", "kriging_tools", "ok", "ok3d", "rk", "uk", "uk3d", "variogram_models"):
    sys.modules["pykrige." + _n] = getattr(pykrige_amd, _n)
'''

# Tests of the reference's suite that are NOT expected to pass against the drop-in, each with its reason.  SURVEY 8(a) lists the
# reference behaviours a replacement should not reproduce; the rest are arguments this design does not have.
EXPECTED_NOT_PASSING = {
}


def _run(tmp_path, files):
    with zipfile.ZipFile(TESTS_ZIP) as z:
        z.extractall(tmp_path)
    (tmp_path / "conftest.py").write_text(SHIM % ROOT)
    xml = tmp_path / "report.xml"
    env = dict(os.environ, PYTHONPATH=ROOT, MPLBACKEND="Agg")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-p", "no:cacheprovider", "--junitxml", str(xml), "-o", "junit_family=xunit1"]
                       + ["tests/" + f for f in files], cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=1500)
    assert xml.exists(), (r.stdout[-3000:], r.stderr[-3000:])
    out = {}
    for case in ET.parse(str(xml)).getroot().iter("testcase"):
        name = case.get("name")
        kind = "passed"
        detail = ""
        for child in case:
            if child.tag in ("failure", "error"):
                kind, detail = "failed", (child.get("message") or "")[:300]
            elif child.tag == "skipped":
                kind, detail = "skipped", (child.get("message") or "")[:200]
        out[name] = (kind, detail)
    return out, r.stdout[-4000:]


@pytest.mark.gpu
def test_reference_suite_passes_against_the_drop_in(tmp_path):
    if not os.path.exists(TESTS_ZIP):
        pytest.skip("oracle/_ref/reference_tests.zip not staged (oracle/build_ref.sh needs /root/reference)")
    res, tail = _run(tmp_path, ["test_core.py", "test_api.py", "test_regression_krige.py", "test_classification_krige.py"])
    passed = sorted(k for k, v in res.items() if v[0] == "passed")
    failed = {k: v[1] for k, v in res.items() if v[0] == "failed"}
    skipped = {k: v[1] for k, v in res.items() if v[0] == "skipped"}
    print("reference suite against pykrige_amd: %d passed, %d failed, %d skipped of %d" % (len(passed), len(failed), len(skipped), len(res)))
    for k, v in sorted(failed.items()):
        print("  FAILED %s: %s" % (k, v))
    for k, v in sorted(skipped.items()):
        print("  skipped %s: %s" % (k, v))
    unexpected = {k: v for k, v in failed.items() if k.split("[")[0] not in EXPECTED_NOT_PASSING}
    assert not unexpected, (unexpected, tail)
    assert len(passed) >= 48, (len(passed), sorted(failed), sorted(skipped))
    # the only tests that may be skipped: GSTools is not installed on the box; the housing data set needs a network
    for k, why in skipped.items():
        assert "gstools" in why or "california housing" in why, (k, why)
