"""Builds pykrige_amd/libmikrige.so (HIP, gfx950) in-tree.  `python -m pykrige_amd.build [--force]`."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "csrc", "mikrige.hip")
DEPS = [SRC, os.path.join(HERE, "csrc", "mik_kernels.h"), os.path.join(ROOT, "include", "mikrige.h")]
OUT = os.path.join(HERE, "libmikrige.so")


def hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (need ROCm; set HIPCC=/path/to/hipcc)")


def up_to_date():
    if not os.path.exists(OUT):
        return False
    t = os.path.getmtime(OUT)
    return all(os.path.getmtime(d) <= t for d in DEPS)


def build_library(force=False, verbose=False):
    """Compile the C-ABI library for gfx950.  Cross-compiles without a GPU."""
    if not force and up_to_date():
        return OUT
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
           "-I" + os.path.join(ROOT, "include"), SRC, "-o", OUT + ".tmp", "-ldl"]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    os.replace(OUT + ".tmp", OUT)
    return OUT


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
