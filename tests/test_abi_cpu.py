"""CPU: librmu.so builds, loads and exports every symbol include/rmu.h declares (no compute calls)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "rmu.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rmu_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported(librmu):
    names = _declared()
    assert len(names) >= 18
    missing = [n for n in names if not hasattr(librmu, n)]
    assert not missing, f"librmu.so does not export {missing}"


def test_binding_lists_every_declared_symbol():
    from ragmeup_amd import _native
    assert sorted(_native.SYMBOLS) == _declared()


def test_version_and_error_strings(librmu):
    assert b"gfx950" in librmu.rmu_version()
    assert isinstance(librmu.rmu_last_error(), bytes)


def test_invalid_arguments_fail_without_gpu(librmu):
    """Argument validation happens before any HIP call, so it is testable on the GPU-less builder."""
    import ctypes
    h = ctypes.c_void_p()
    assert librmu.rmu_index_create(None, 384, 0, 0) == -1
    assert librmu.rmu_index_create(ctypes.byref(h), 0, 0, 0) == -1
    assert librmu.rmu_index_create(ctypes.byref(h), 4096, 0, 0) == -1
    assert librmu.rmu_index_create(ctypes.byref(h), 384, 7, 0) == -1
    assert b"metric" in librmu.rmu_last_error()
    assert librmu.rmu_index_search(None, None, 1, 10, 0, 0, None, None, 0) == -1
    assert librmu.rmu_topk_merge(None, None, 1, 1, 10, 0, None, None, 0) == -1
    v = ctypes.c_double(0.0)
    assert librmu.rmu_probe_mfma_rate(0, 0, 100, None) == -1
    assert librmu.rmu_probe_mfma_rate(2, 0, 100, ctypes.byref(v)) == -1
    assert librmu.rmu_probe_mfma_rate(1, 1, 100, ctypes.byref(v)) == -1       # (the LDS / DMA skeleton is the f16 kernel's)
    assert librmu.rmu_probe_mfma_rate(0, 0, 0, ctypes.byref(v)) == -1


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under ragmeup_amd/ may import, load or exec it."""
    pkg = os.path.join(ROOT, "ragmeup_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f), errors="replace").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f
                assert "liboracle" not in txt and "oracle/" not in txt.replace("the oracle/", ""), f


def test_header_is_plain_c_and_the_c_example_links(tmp_path, librmu):
    """include/rmu.h must be consumable by a C compiler (the boundary is a C ABI, not a C++ one), and a plain-C program
    must link against librmu.so with nothing but -lrmu (no execution here: the library needs a GPU)."""
    import os
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gcc = shutil.which("gcc")
    assert gcc, "gcc is part of the build image"
    probe = tmp_path / "probe.c"
    probe.write_text('#include "rmu.h"\nint main(void) { return rmu_version() == 0; }\n')
    subprocess.run([gcc, "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-I", os.path.join(root, "include"), str(probe)],
                   check=True, capture_output=True)
    libdir = os.path.join(root, "ragmeup_amd", "lib")
    exe = tmp_path / "flat_search"
    r = subprocess.run([gcc, "-std=c99", "-O1", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(root, "include"),
                        os.path.join(root, "examples", "flat_search.c"), "-L", libdir, "-lrmu", "-Wl,-rpath," + libdir,
                        "-Wl,--allow-shlib-undefined", "-lm", "-o", str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert exe.exists()


def test_the_library_reads_tuning_switches_only_behind_the_master_switch():
    """Every RMU_* environment switch of librmu goes through rmu_env(), which answers only when RMU_TUNING=1 is set: a stray variable in
    a server's environment cannot change which kernels run."""
    import glob
    import re
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ragmeup_amd", "csrc")
    direct = []
    for f in glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h")):
        for m in re.finditer(r'(?<![_a-z])getenv\("(\w+)"\)', open(f).read()):
            if m.group(1) != "RMU_TUNING":
                direct.append((os.path.basename(f), m.group(1)))
    assert direct == []
    common = open(os.path.join(csrc, "rmu_common.h")).read()
    assert 'inline const char* rmu_env(const char* name)' in common and 'getenv("RMU_TUNING")' in common


def test_the_product_library_carries_only_the_kernels_it_takes():
    """the superseded forms of the screening kernel (scan_screen_kernel, scan_screen_lean_kernel,
    scan_screen_lean2_kernel, the K-split / 128-queries-per-wave experiments) and the round-1/2 encoder kernels are instantiated in debug
    builds only; the product library's symbol table names scan_screen_lean3_kernel (three geometries, each in its inner-product and its
    L2 form -- round 5 -- and, round 6, each of those with the 128-key candidate slots of 32 < k <= 104) and nothing else of that family."""
    import re
    import shutil
    import subprocess
    from ragmeup_amd import _native
    nm = shutil.which("nm") or "/opt/rocm/lib/llvm/bin/llvm-nm"
    out = subprocess.run([nm, "-C", _native.SO_PATH], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    screen = sorted(set(re.findall(r"scan_screen\w*kernel<[^>]*>", out.stdout)))
    assert screen == [f"scan_screen_lean3_kernel<0, {g}, {l2}, {deep}>" for g in ("4, 0", "4, 1", "8, 0") for l2 in (0, 1) for deep in ("false", "true")], screen
    for gone in ("k_ffn_fused", "k_attention<", "k_attn4", "k_gemm_mid"):
        assert gone not in out.stdout, gone



def test_no_device_wide_synchronisation_and_every_entry_point_in_relaxed_capture_mode():
    """(round 6; profiles/r06_capture_probe.txt) A hipDeviceSynchronize, a synchronous hipMemcpy / hipMemset on ANY thread invalidates a
    hipGraph capture running on another thread -- the reference's LLM captures in the same process (server/RAGHelper_local.py:42-105) --
    and a thread in the default capture-interaction mode does so with most synchronous calls.  The library's product sources therefore
    (1) never call them (the debug-counter dumps under RMU_DEBUG_KERNELS / RMU_SCAN_EXP aside) and (2) open every exported entry point
    that reaches HIP with RMU_ENTRY() (relaxed mode for the duration of the call); its own captures are still taken one at a time."""
    import glob
    import os
    import re
    src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ragmeup_amd", "csrc")
    bare = []
    for path in sorted(glob.glob(os.path.join(src, "*.hip")) + glob.glob(os.path.join(src, "*.h"))):
        text = open(path, encoding="utf-8").read()
        # debug-only regions: everything between #ifdef RMU_DEBUG_KERNELS and its #endif / #else, and the RMU_SCAN_EXP counter dump
        text = re.sub(r"#ifdef RMU_DEBUG_KERNELS.*?#(?:endif|else)", lambda m: "\n" * m.group(0).count("\n"), text, flags=re.S)
        for i, ln in enumerate(text.split("\n")):
            code = ln.split("//")[0]
            if re.search(r"\bhipDeviceSynchronize\(|\bhipMemcpy\(|\bhipMemset\(|\bhipMemcpy2D\(", code) and "dbg" not in code:
                bare.append(f"{os.path.basename(path)}:{i + 1}: {code.strip()[:80]}")
    assert not bare, bare
    common = open(os.path.join(src, "rmu_common.h"), encoding="utf-8").read()
    assert "hipThreadExchangeStreamCaptureMode" in common and "#define RMU_ENTRY()" in common
    # every `extern "C" int rmu_*` with a body of its own lines that touches HIP starts with RMU_ENTRY()
    missing = []
    for name in ("rmu_api.hip", "bert.hip", "rmu_comm.hip"):
        text = open(os.path.join(src, name), encoding="utf-8").read()
        for m in re.finditer(r'^extern "C" int (rmu_\w+)\([^;{]*\)\s*\{\n(.*?)^\}', text, flags=re.S | re.M):
            fn, body = m.group(1), m.group(2)
            if re.search(r"\bhip[A-Z]\w+\(|ensure_stream|hipLaunchKernelGGL|nccl[A-Z]", body) and "RMU_ENTRY();" not in body.split("\n")[0]:
                missing.append(fn)
    assert missing == [], missing
    bert = open(os.path.join(src, "bert.hip"), encoding="utf-8").read()
    cap = bert.index("hipStreamBeginCapture(")
    assert "rmu_capture_mutex()" in bert[cap - 400:cap]
    assert re.search(r"inline std::mutex& rmu_capture_mutex\(\)", common)


def test_no_kernel_of_the_product_library_spills_to_scratch(tmp_path):
    """(round 6) Two uniform `if`s added to k_ffn3 for an A/B experiment cost the register allocator 196 bytes of scratch per lane: +4.8 GB of
    HBM traffic per forward and 3.05 -> 3.30 ms per launch, found only because a PMC table looked wrong.  The code objects inside librmu.so
    carry every kernel's resources: none of the hot kernels may have a private (scratch) segment."""
    import re
    import shutil
    import subprocess
    from ragmeup_amd import _native
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    readelf = "/opt/rocm/lib/llvm/bin/llvm-readelf"
    if not (os.path.exists(objdump) and os.path.exists(readelf)):
        import pytest
        pytest.skip("ROCm's llvm-objdump / llvm-readelf are not installed")
    so = tmp_path / "librmu.so"
    # the PRODUCT library, whatever RMU_LIB selects for this process (`make asan-test` runs the suite against librmu_asan.so, whose device code is
    # compiled with -g and the sanitizer flags: its register allocation is not the shipped one -- the 768-wide exact-scan instantiations keep 20-36
    # bytes of scratch there)
    product = os.path.join(os.path.dirname(_native.__file__), "lib", "librmu.so")
    shutil.copy(product if os.path.exists(product) else _native.SO_PATH, so)
    r = subprocess.run([objdump, "--offloading", str(so)], capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 0, r.stderr[-2000:]
    found, spilled = 0, []
    for co in sorted(tmp_path.glob("librmu.so.*gfx950")):
        notes = subprocess.run([readelf, "--notes", str(co)], capture_output=True, text=True).stdout
        for name, scratch in re.findall(r"\.name:\s+(\S+)\n\s+\.private_segment_fixed_size:\s+(\d+)", notes):
            found += 1
            if int(scratch) != 0:
                spilled.append((name[:70], int(scratch)))
    assert found > 40, found                                   # the metadata was really read
    hot = [s for s in spilled if re.search(r"k_ffn3|k_gemm3|k_attn3|k_gemm|scan_screen_lean3|scan_topk|k_qa|k_rescore|merge_select", s[0])]
    assert hot == [], hot
