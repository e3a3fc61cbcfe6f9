"""TEST / BASELINE INFRASTRUCTURE.  The REAL reference as an importable package where /root/reference does not exist (the GPU box):
oracle/build_ref.sh stages its pure-Python modules as oracle/_ref/pykrige_py.zip (zipimport) and its two compiled extensions in
oracle/_ref/pykrige_lib/.  `import_reference()` returns the `pykrige` package itself -- ok.py, uk.py, ok3d.py, uk3d.py, core.py as
written upstream, with `backend='C'` wired to the compiled loops -- so that bench.py's cpu_baseline leg times the reference, not a
restatement of it (kind "reference"), and tests can compare against it on the GPU box.  Only tests/, bench.py's cpu_baseline leg
and __graft_entry__ may import this module; pykrige_amd never does."""
import glob
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
_ZIP = os.path.join(_HERE, "_ref", "pykrige_py.zip")
_LIB = os.path.join(_HERE, "_ref", "pykrige_lib")
TESTS_ZIP = os.path.join(_HERE, "_ref", "reference_tests.zip")


def available():
    return os.path.exists(_ZIP)


def c_available():
    return bool(glob.glob(os.path.join(_LIB, "cok*.so")))


def import_reference(stub_statistics=True):
    """The reference package (from the staged archive).  stub_statistics: replace the O(N^4) constructor-time cross-validation
    statistics (`_find_statistics`, run unconditionally by three constructors -- uk.py:380, ok3d.py:352, uk3d.py:380)
    by a stub, as BASELINE.md section 3 prescribes for timing execute()."""
    if not available():
        raise ImportError("oracle/_ref/pykrige_py.zip not staged (run oracle/build_ref.sh where /root/reference exists)")
    mod = sys.modules.get("pykrige")
    if mod is not None and not getattr(mod, "__file__", None):  # the bare namespace oracle/ref_c_loop.py registers for the .so pair
        for k in [k for k in sys.modules if k == "pykrige" or k.startswith("pykrige.")]:
            if k not in ("pykrige.lib.cok", "pykrige.lib.variogram_models"):
                del sys.modules[k]
    if _ZIP not in sys.path:
        sys.path.insert(0, _ZIP)
    import pykrige  # noqa: E402
    import pykrige.lib  # noqa: E402

    if c_available() and _LIB not in list(pykrige.lib.__path__):
        pykrige.lib.__path__.append(_LIB)
    import pykrige.ok  # noqa: E402
    import pykrige.ok3d  # noqa: E402
    import pykrige.uk  # noqa: E402
    import pykrige.uk3d  # noqa: E402

    if stub_statistics:
        import numpy as np

        def stub(*a, **k):
            return np.zeros(2), np.ones(2), np.zeros(2)

        for m in (pykrige.ok, pykrige.uk, pykrige.ok3d, pykrige.uk3d):
            m._find_statistics = stub
    return pykrige
