"""Builds pykrige_amd/libmikrige.so (HIP, gfx950) in-tree.  `python -m pykrige_amd.build [--force] [-j N]`.

The library is eight translation units compiled in parallel (objects under pykrige_amd/csrc/build/, git- and gpurun-ignored) and
linked into one shared object; only the units whose sources changed are recompiled:

    mikrige.hip       C ABI, handles, device groups + factor exchange, K1 assembly, points / grids / masks, statistics
    mik_inverse.hip   K2: block Gauss-Jordan sweep and its schedules, probes, pseudo-inverses
    mik_predict.hip   K3: right-hand sides, dense and range-aware contraction, point sort
    mik_mw.hip        moving window: neighbour search, Gauss-Jordan / blocked / HBM solvers, dispatch
    mik_mw_chol.hip   x 4 (-DMIK_MWC_PART=0..3): the register-tile classes of the moving window's LDL^T solver
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
OUT = os.path.join(HERE, "libmikrige.so")
API = os.path.join(ROOT, "include", "mikrige.h")
COMMON = ["mik_host.h", "mik_dev.h"]
# object name -> (source, extra headers it includes, extra flags)
UNITS = {
    "mikrige": ("mikrige.hip", ["mik_k_core.h"], []),
    "mik_inverse": ("mik_inverse.hip", ["mik_k_inverse.h"], []),
    "mik_predict": ("mik_predict.hip", ["mik_k_predict.h"], []),
    "mik_mw": ("mik_mw.hip", ["mik_k_mw.h", "mik_k_mw_chol.h"], []),
    "mik_mw_chol0": ("mik_mw_chol.hip", ["mik_k_mw_chol.h"], ["-DMIK_MWC_PART=0"]),
    "mik_mw_chol1": ("mik_mw_chol.hip", ["mik_k_mw_chol.h"], ["-DMIK_MWC_PART=1"]),
    "mik_mw_chol2": ("mik_mw_chol.hip", ["mik_k_mw_chol.h"], ["-DMIK_MWC_PART=2"]),
    "mik_mw_chol3": ("mik_mw_chol.hip", ["mik_k_mw_chol.h"], ["-DMIK_MWC_PART=3"]),
}
# --offload-compress: the code objects travel zstd-compressed inside the fat binary (16.6 -> 5 MB of .so; the HIP runtime unpacks them at load)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "--offload-compress", "-I" + os.path.join(ROOT, "include")]


def hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (need ROCm; set HIPCC=/path/to/hipcc)")


def unit_deps(name):
    src, hdrs, _ = UNITS[name]
    return [os.path.join(CSRC, f) for f in [src] + hdrs + COMMON] + [API]


def all_sources():
    seen = []
    for name in UNITS:
        for d in unit_deps(name):
            if d not in seen:
                seen.append(d)
    return seen


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def up_to_date():
    return not _stale(OUT, all_sources())


def build_library(force=False, verbose=False, jobs=None):
    """Compile the C-ABI library for gfx950.  Cross-compiles without a GPU."""
    if not force and up_to_date():
        return OUT
    os.makedirs(OBJ, exist_ok=True)
    cc = hipcc()
    todo = []
    for name, (src, _, extra) in UNITS.items():
        obj = os.path.join(OBJ, name + ".o")
        if force or _stale(obj, unit_deps(name)):
            todo.append((name, [cc] + FLAGS + extra + ["-c", os.path.join(CSRC, src), "-o", obj]))

    def run(job):
        name, cmd = job
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0 and "--offload-compress" in cmd and "offload-compress" in r.stderr:
            # a hipcc that does not know the flag (before ROCm 6.1): the same objects, uncompressed
            r = subprocess.run([c for c in cmd if c != "--offload-compress"], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("compiling %s failed:\n%s" % (name, r.stderr[-4000:]))
        return name

    # the four mik_mw_chol units are the long ones: start them first
    todo.sort(key=lambda j: 0 if j[0].startswith("mik_mw_chol") else 1)
    with ThreadPoolExecutor(max_workers=jobs or min(len(UNITS), os.cpu_count() or 4)) as ex:
        list(ex.map(run, todo))
    link = [cc, "--offload-arch=gfx950", "-shared", "-fPIC"] + [os.path.join(OBJ, n + ".o") for n in UNITS] + ["-o", OUT + ".tmp", "-ldl"]
    if verbose:
        print(" ".join(link), flush=True)
    subprocess.run(link, check=True)
    os.replace(OUT + ".tmp", OUT)
    return OUT


if __name__ == "__main__":
    j = int(sys.argv[sys.argv.index("-j") + 1]) if "-j" in sys.argv else None
    print(build_library(force="--force" in sys.argv, verbose=True, jobs=j))
