"""Build libdhr_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import shutil
import subprocess

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libdhr_hip.so")
SOURCES = ["kernels.hip", "gemm_w4.hip", "api.hip", "sharded.hip", "pq_adc.hip", "select_global.hip", "host_io.hip"]
HEADERS = ["dhr_internal.h", "gemm_common.h", os.path.join("..", "..", "include", "dhr_hip.h")]


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value", "-Wno-unused-result",
           "-o", LIB] + SOURCES + ["-L/opt/rocm/lib", "-lrccl"]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, cwd=CSRC, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
