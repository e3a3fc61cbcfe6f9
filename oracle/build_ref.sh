#!/bin/bash
# TEST INFRASTRUCTURE.  Stages the REAL reference next to the repo's oracle, from the sources WHERE THEY LIE under /root/reference,
# with all intermediates in a temp dir and only BUILD OUTPUTS in oracle/_ref/ (git-ignored; it travels to the GPU box with the
# in-tree .so files, /root/reference does not exist there):
#   oracle/_ref/pykrige_lib/{cok,variogram_models}*.so   the reference's two Cython extensions (lib/cok.pyx, lib/variogram_models.pyx),
#                                                        built as /root/reference/setup.py:10-21 does
#   oracle/_ref/pykrige_py.zip                           the reference's pure-Python package (src/pykrige/*.py), importable through
#                                                        zipimport (oracle/ref_package.py): the CPU baseline times the reference ITSELF
#   oracle/_ref/reference_tests.zip                      the reference's own test-suite (tests/*.py + tests/test_data/): run against
#                                                        the drop-in by tests/test_reference_suite.py
# Nothing of the reference enters the repository's history; archives are unpacked only into pytest's tmp_path at test time.
# The reference's own pyproject build needs setuptools>=77/setuptools_scm/pentapy (absent offline), hence this recipe.
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
REFROOT=/root/reference
REF=$REFROOT/src/pykrige/lib
[ -d "$REF" ] || { echo "no reference here; nothing to build"; exit 0; }
OUT="$HERE/_ref/pykrige_lib"; mkdir -p "$OUT"
TMP="$(mktemp -d)"; trap 'rm -rf "$TMP"' EXIT
cd "$TMP"
INC=$(python3 -c 'import numpy, sysconfig; print("-I"+numpy.get_include(), "-I"+sysconfig.get_paths()["include"])')
SUF=$(python3 -c 'import sysconfig; print(sysconfig.get_config_var("EXT_SUFFIX"))')
for m in variogram_models cok; do
  python3 -m cython -3 -I "$REF" "$REF/$m.pyx" -o "$TMP/$m.c"
  gcc -O2 -fPIC -shared -DNPY_NO_DEPRECATED_API=NPY_1_7_API_VERSION $INC "$TMP/$m.c" -o "$OUT/$m$SUF"
done
python3 - "$REFROOT" "$HERE/_ref" <<'PY'
import os, sys, zipfile
ref, out = sys.argv[1], sys.argv[2]
with zipfile.ZipFile(os.path.join(out, "pykrige_py.zip"), "w", zipfile.ZIP_DEFLATED) as z:
    src = os.path.join(ref, "src")
    for d, _, files in os.walk(os.path.join(src, "pykrige")):
        for f in sorted(files):
            if f.endswith(".py"):
                p = os.path.join(d, f)
                z.write(p, os.path.relpath(p, src))
with zipfile.ZipFile(os.path.join(out, "reference_tests.zip"), "w", zipfile.ZIP_DEFLATED) as z:
    tests = os.path.join(ref, "tests")
    for d, _, files in os.walk(tests):
        if "__pycache__" in d:
            continue
        for f in sorted(files):
            p = os.path.join(d, f)
            z.write(p, os.path.join("tests", os.path.relpath(p, tests)))
PY
echo "built: $(ls $OUT) $(ls $HERE/_ref/*.zip)"
