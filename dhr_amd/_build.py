"""Build libdhr_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

One process per GPU is the product's launch mode, so several ranks may find the library stale at the same moment: the
stale check and the build run under an flock, objects and the library are written to temporary names and renamed into
place (a concurrent CDLL never sees a half-written ELF)."""
from __future__ import annotations

import concurrent.futures
import fcntl
import os
import shutil
import subprocess

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libdhr_hip.so")
OBJDIR = os.path.join(CSRC, "build")
SOURCES = ["abi.cpp", "kernels.hip", "gemm_w4.hip", "gemm_g8.hip", "index_build.hip", "search_core.hip", "api.hip", "sharded.hip", "pq_adc.hip", "select_global.hip", "host_io.hip"]
HEADERS = ["dhr_internal.h", "dhr_state.h", "abi_guard.h", "libdhr.map", "gemm_common.h", "gemm_g8.h", os.path.join("..", "..", "include", "dhr_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-Wno-unused-result"]
# A/B builds only (DHR_AB_VARIANTS=1 in the environment of the build, tools/ab_build.sh): the retired persistent-workgroup form of the integer
# bound GEMM (tools/ab/gemm_g8p.hip, DHR_PARAM_GEMM_VARIANT = 6) -- the measurement behind DESIGN.md section 4b, not part of the shipped library
if os.environ.get("DHR_AB_VARIANTS") == "1":
    SOURCES = SOURCES + [os.path.join("..", "..", "tools", "ab", "gemm_g8p.hip")]
    FLAGS = FLAGS + ["-DDHR_AB_VARIANTS", "-I" + CSRC]


def _mtime(path: str) -> float:
    return os.path.getmtime(path) if os.path.exists(path) else 0.0


def _stale() -> bool:
    t = _mtime(LIB)
    return t == 0.0 or any(_mtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def _compile(hipcc: str, src: str, newest_header: float, force: bool, verbose: bool) -> str:
    obj = os.path.join(OBJDIR, os.path.splitext(os.path.basename(src))[0] + ".o")
    if not force and _mtime(obj) > max(_mtime(os.path.join(CSRC, src)), newest_header):
        return obj
    tmp = obj + ".tmp.%d" % os.getpid()
    cmd = [hipcc] + FLAGS + ["-c", src, "-o", tmp]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, cwd=CSRC, check=True)
    os.replace(tmp, obj)
    return obj


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return LIB
    os.makedirs(OBJDIR, exist_ok=True)
    with open(os.path.join(OBJDIR, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not _stale():          # another rank built it while this one waited
                return LIB
            hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
            newest_header = max(_mtime(os.path.join(CSRC, h)) for h in HEADERS)
            with concurrent.futures.ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as ex:
                objs = list(ex.map(lambda s: _compile(hipcc, s, newest_header, force, verbose), SOURCES))
            tmp = LIB + ".tmp.%d" % os.getpid()
            cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--version-script=" + os.path.join(CSRC, "libdhr.map"), "-o", tmp] + objs + ["-L/opt/rocm/lib", "-lrccl"]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, cwd=CSRC, check=True)
            os.replace(tmp, LIB)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB


if __name__ == "__main__":
    import sys
    print(build(force="--force" in sys.argv, verbose=True))
