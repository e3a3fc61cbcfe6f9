"""Builds pykrige_amd/libmikrige.so (HIP, gfx950) in-tree.  `python -m pykrige_amd.build [--force] [-j N]`.

The library is ten translation units compiled in parallel (objects under pykrige_amd/csrc/build/, git- and gpurun-ignored) and
linked into one shared object; only the units whose sources or flags changed are recompiled (the command line of every object is
kept beside it as <name>.flags).  Safe against concurrent callers (pytest-xdist, one process per GPU calling __graft_entry__.build):
the whole build holds an flock on csrc/build/.lock, and objects / the library are written under a per-process name and renamed into place.

    mikrige.hip       C ABI, handles, device groups + factor exchange, K1 assembly, points / grids / masks, statistics
    mik_inverse.hip   K2: block Gauss-Jordan sweep and its schedules, probes, pseudo-inverses
    mik_predict.hip   K3: right-hand sides, dense and range-aware contraction, point sort
    mik_mw.hip        moving window: neighbour search, blocked / HBM solvers, dispatch
    mik_mw_solve.hip  moving window: the register classes of the pivoting Gauss-Jordan solver (the LDL^T kernels' fallback)
    mik_mw_chol.hip   x 5 (-DMIK_MWC_PART=0..4): the register-tile classes of the moving window's LDL^T solver
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
    "mik_mw_solve": ("mik_mw_solve.hip", ["mik_k_mw_solve.h", "mik_k_mw_chol.h"], []),
    "mik_mw_chol0": ("mik_mw_chol.hip", ["mik_k_mw_chol.h"], ["-DMIK_MWC_PART=0"]),
    "mik_mw_chol1": ("mik_mw_chol.hip", ["mik_k_mw_chol.h"], ["-DMIK_MWC_PART=1"]),
    "mik_mw_chol2": ("mik_mw_chol.hip", ["mik_k_mw_chol.h"], ["-DMIK_MWC_PART=2"]),
    "mik_mw_chol3": ("mik_mw_chol.hip", ["mik_k_mw_chol.h"], ["-DMIK_MWC_PART=3"]),
    "mik_mw_chol4": ("mik_mw_chol.hip", ["mik_k_mw_chol.h"], ["-DMIK_MWC_PART=4"]),
}
# --offload-compress: the code objects travel zstd-compressed inside the fat binary (16.9 -> 2.3 MB of .so; the HIP runtime unpacks them at load);
# level 19 instead of the default 3 takes another 3 % off and no measurable compile time (the time is in code generation)
COMPRESS = ["--offload-compress", "--offload-compression-level=19"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"] + COMPRESS + ["-I" + os.path.join(ROOT, "include")]
# seconds of one core per unit (hipcc of ROCm 7.2): the order the units are started in, longest first
COST = {"mik_mw_chol": 23, "mik_mw_solve": 15, "mik_predict": 12, "mikrige": 9, "mik_inverse": 6, "mik_mw": 4}


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


def _stale(target, deps, flags=None):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    if any(os.path.getmtime(d) > t for d in deps):
        return True
    if flags is not None:  # built with another command line (e.g. the fallback without --offload-compress): stale
        try:
            return open(target[:-2] + ".flags").read() != " ".join(flags)
        except OSError:
            return True
    return False


def up_to_date():
    return not _stale(OUT, all_sources())


def build_library(force=False, verbose=False, jobs=None):
    """Compile the C-ABI library for gfx950.  Cross-compiles without a GPU."""
    if not force and up_to_date():
        return OUT
    os.makedirs(OBJ, exist_ok=True)
    import fcntl

    with open(os.path.join(OBJ, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)  # one builder at a time; the others find the work done when they get the lock
        if not force and up_to_date():
            return OUT
        return _build_locked(force, verbose, jobs)


def _build_locked(force, verbose, jobs):
    cc = hipcc()
    pid = ".%d.tmp" % os.getpid()
    todo = []
    for name, (src, _, extra) in UNITS.items():
        obj = os.path.join(OBJ, name + ".o")
        flags = FLAGS + extra
        if force or _stale(obj, unit_deps(name), flags):
            todo.append((name, flags, [cc] + flags + ["-c", os.path.join(CSRC, src), "-o", obj + pid]))

    def run(job):
        name, flags, cmd = job
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0 and COMPRESS[1] in cmd and "offload-compress" in r.stderr:
            # a hipcc that knows --offload-compress but not the level: the default level
            flags = [c for c in flags if c != COMPRESS[1]]
            cmd = [c for c in cmd if c != COMPRESS[1]]
            r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0 and COMPRESS[0] in cmd and "offload-compress" in r.stderr:
            # a hipcc that does not know the flag (before ROCm 6.1): the same objects, uncompressed -- and recorded as such, so that
            # a later build with a newer hipcc sees them as stale (the library is several times larger without the compression)
            flags = [c for c in flags if c not in COMPRESS]
            print("pykrige_amd.build: this hipcc does not know --offload-compress; %s is built uncompressed" % name, file=sys.stderr)
            r = subprocess.run([c for c in cmd if c not in COMPRESS], capture_output=True, text=True)
        obj = os.path.join(OBJ, name + ".o")
        if r.returncode != 0:
            if os.path.exists(obj + pid):
                os.remove(obj + pid)
            raise RuntimeError("compiling %s failed:\n%s" % (name, r.stderr[-4000:]))
        os.replace(obj + pid, obj)
        with open(obj[:-2] + ".flags", "w") as f:
            f.write(" ".join(flags))
        return name

    todo.sort(key=lambda j: -COST.get(j[0].rstrip("0123456789"), 0))
    with ThreadPoolExecutor(max_workers=jobs or min(len(UNITS), os.cpu_count() or 4)) as ex:
        list(ex.map(run, todo))
    link = [cc, "--offload-arch=gfx950", "-shared", "-fPIC"] + [os.path.join(OBJ, n + ".o") for n in UNITS] + ["-o", OUT + pid, "-ldl", "-Wl,-s"]
    # (-s: no static symbol table -- 90 kB; the C ABI and the kernels' host stubs are dynamic symbols, `nm -D` lists them)
    if verbose:
        print(" ".join(link), flush=True)
    subprocess.run(link, check=True)
    os.replace(OUT + pid, OUT)
    return OUT


if __name__ == "__main__":
    j = int(sys.argv[sys.argv.index("-j") + 1]) if "-j" in sys.argv else None
    print(build_library(force="--force" in sys.argv, verbose=True, jobs=j))
