"""Build librmu.so (hipcc, gfx950 only) in-tree: ragmeup_amd/lib/librmu.so.

hipcc cross-compiles without a GPU; the built .so travels to the GPU box with the repo snapshot
(git-ignored, not gpurun-ignored).
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(HERE, "lib", "obj")
SO = os.path.join(LIBDIR, "librmu.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def _sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs.append(os.path.join(HERE, "..", "include", "rmu.h"))
    return hs


def _stale(target: str, deps: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False, debug_kernels: bool = False) -> str:
    """debug_kernels: also compile the cycle-counter / ablation instantiations (-DRMU_DEBUG_KERNELS: RMU_FFN_DBG, RMU_G3_DBG,
    RMU_GEMM_DBG); the product library carries none of them.  Switching the flag needs force=True."""
    os.makedirs(OBJDIR, exist_ok=True)
    srcs = _sources()
    hdrs = _headers()
    objs = []
    jobs = []
    for s in srcs:
        o = os.path.join(OBJDIR, os.path.basename(s)[:-4] + ".o")
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            jobs.append([HIPCC, *FLAGS, *(["-DRMU_DEBUG_KERNELS"] if debug_kernels else []), "-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        return r

    if jobs:
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if force or jobs or _stale(SO, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO, *objs])
    return SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv or "--debug-kernels" in sys.argv, verbose=True, debug_kernels="--debug-kernels" in sys.argv))
