#!/bin/bash
# TEST INFRASTRUCTURE.  Builds the reference's two Cython extensions (lib/cok.pyx, lib/variogram_models.pyx)
# from the sources WHERE THEY LIE under /root/reference, mirroring /root/reference/setup.py:10-21, with all
# intermediates in a temp dir and only the .so outputs in oracle/_ref/pykrige_lib/ (git-ignored).
# The reference's own pyproject build needs setuptools>=77/setuptools_scm/pentapy (absent offline), hence this recipe.
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
REF=/root/reference/src/pykrige/lib
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
echo "built: $(ls $OUT)"
