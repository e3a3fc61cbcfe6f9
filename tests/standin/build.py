"""Builds the RCCL stand-ins of the exchange tests next to this file (test infrastructure; see standin_rccl.cpp)."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "standin_rccl.cpp")
CPU = os.path.join(HERE, "libstandin_rccl_cpu.so")
HIP = os.path.join(HERE, "libstandin_rccl_hip.so")


def _stale(out):
    return not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(SRC)


def build(hip=True):
    if _stale(CPU):
        subprocess.run(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", SRC, "-o", CPU, "-lpthread"], check=True)
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if hip and os.path.exists(hipcc) and _stale(HIP):
        subprocess.run([hipcc, "-O1", "-std=c++17", "-shared", "-fPIC", "-DWITH_HIP", "-x", "hip", "--offload-arch=gfx950", SRC, "-o", HIP],
                       check=True)
    return CPU, HIP


if __name__ == "__main__":
    print(build())
