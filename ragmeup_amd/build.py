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


SAN_FLAGS = ["-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-g", "-shared-libsan"]


def build(force: bool = False, verbose: bool = False, debug_kernels: bool = False, asan: bool = False) -> str:
    """debug_kernels: also compile the cycle-counter / ablation instantiations (-DRMU_DEBUG_KERNELS: RMU_FFN_DBG, RMU_G3_DBG,
    RMU_GEMM_DBG); the product library carries none of them.  Switching the flag needs force=True.
    asan: a SEPARATE library, lib/librmu_asan.so (objects under lib/obj_asan), whose HOST code -- the C-ABI, the WordPiece tokenizer and
    its thread pool, the index's locking and bookkeeping -- is compiled with AddressSanitizer + UBSan (`make asan-test` runs the CPU
    tests that call into the library against it; device code is not instrumented: GPU sanitizers are not available on this pool)."""
    # (round 6) a debug-kernels build is a separate library as well (lib/librmu_dbg.so): RMU_TUNING=1 RMU_LIB=<that> selects it for one
    # process, the product library stays in place
    objdir = os.path.join(LIBDIR, "obj_asan") if asan else os.path.join(LIBDIR, "obj_dbg") if debug_kernels else OBJDIR
    so = os.path.join(LIBDIR, "librmu_asan.so") if asan else os.path.join(LIBDIR, "librmu_dbg.so") if debug_kernels else SO
    os.makedirs(objdir, exist_ok=True)
    srcs = _sources()
    hdrs = _headers()
    objs = []
    jobs = []
    for s in srcs:
        o = os.path.join(objdir, os.path.basename(s)[:-4] + ".o")
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            jobs.append([HIPCC, *FLAGS, *(SAN_FLAGS if asan else []), *(["-DRMU_DEBUG_KERNELS"] if debug_kernels else []), "-c", s, "-o", o])

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
    if force or jobs or _stale(so, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *(SAN_FLAGS if asan else []), "-o", so, *objs])
    return so


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, debug_kernels="--debug-kernels" in sys.argv,
                asan="--asan" in sys.argv))
