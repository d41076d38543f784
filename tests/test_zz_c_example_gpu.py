"""GPU: the plain-C example of the C-ABI (examples/flat_search.c) is compiled with gcc, linked against librmu.so and RUN --
no Python, no torch in that process.  (Named to sort last: it is the only test that leaves the Python process.)"""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_c_example_runs_against_the_library(tmp_path, librmu):
    gcc = shutil.which("gcc")
    assert gcc
    libdir = os.path.join(ROOT, "ragmeup_amd", "lib")
    exe = tmp_path / "flat_search"
    subprocess.run([gcc, "-std=c99", "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "flat_search.c"),
                    "-L", libdir, "-lrmu", "-Wl,-rpath," + libdir, "-Wl,--allow-shlib-undefined", "-lm", "-o", str(exe)],
                   check=True, capture_output=True)
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = "/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, (r.stdout, r.stderr)
    assert "equals the single-index search: yes" in r.stdout
    for i in range(4):
        assert f"query {i}: best row {i * 777} " in r.stdout, r.stdout
