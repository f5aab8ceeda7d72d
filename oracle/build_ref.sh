#!/usr/bin/env bash
# Builds the reference's only native component (monotonic_align/core.pyx) from the
# sources where they lie under /root/reference into oracle/_ref/ (git-ignored).
# Test infrastructure only: used to pin oracle/mas_ref.c and as cpu_baseline kind "reference".
# Nothing is copied into the repository; outputs (generated C, .so) stay in oracle/_ref/.
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
REF="${GLOWTTS_REFERENCE:-/root/reference}"
OUT="$HERE/_ref"
if [ ! -f "$REF/monotonic_align/core.pyx" ]; then
  echo "[oracle/build_ref] reference not present at $REF - skipping (prebuilt files, if any, are kept)"
  exit 0
fi
mkdir -p "$OUT/monotonic_align"
PYINC=$(python -c "import sysconfig; print(sysconfig.get_paths()['include'])")
NPINC=$(python -c "import numpy; print(numpy.get_include())")
EXT=$(python -c "import sysconfig; print(sysconfig.get_config_var('EXT_SUFFIX'))")
python -m cython -3 "$REF/monotonic_align/core.pyx" -o "$OUT/monotonic_align/core.c"
# same flags as a default distutils build of the reference's setup.py (no -fopenmp: setup.py:5-9 passes none)
gcc -O2 -fPIC -shared -fwrapv -fno-strict-aliasing -DNPY_NO_DEPRECATED_API=NPY_1_7_API_VERSION \
    -I"$PYINC" -I"$NPINC" "$OUT/monotonic_align/core.c" -o "$OUT/monotonic_align/core$EXT"
touch "$OUT/monotonic_align/__init__.py"
echo "[oracle/build_ref] built $OUT/monotonic_align/core$EXT"
