"""Static resource table of every kernel in librmu.so (no GPU needed): compiles each .hip for gfx950 to assembly and reads
the code-object metadata -> profiles/<round>_kernel_resources.md (python tools/kernel_resources.py r03) (arch VGPRs = unified count minus AGPRs, AGPRs, SGPRs, scratch bytes per lane, static LDS).
Dynamic LDS (the scan kernels' rings and candidate slots) is a launch parameter: see rmu_last_scan_geometry / DESIGN.md."""
import glob
import os
import re
import subprocess
import sys
import tempfile

import yaml

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
rows = []
for src in sorted(glob.glob(os.path.join(root, "ragmeup_amd", "csrc", "*.hip"))):
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", out, src],
                           capture_output=True, text=True)
        if r.returncode != 0 or not os.path.exists(out):
            continue                       # host-only translation units (wordpiece.hip) have no device code
        txt = open(out).read()
    block = txt[txt.index(".amdgpu_metadata") + len(".amdgpu_metadata"):txt.index(".end_amdgpu_metadata")]
    meta = yaml.safe_load("\n".join(l for l in block.splitlines() if l.strip() and not l.startswith("\t")))
    for k in meta.get("amdhsa.kernels", []):
        dem = subprocess.run(["c++filt", k[".name"]], capture_output=True, text=True).stdout.strip()
        dem = dem.replace("(anonymous namespace)::", "").replace("void ", "")
        tot, ag = int(k.get(".vgpr_count", 0)), int(k.get(".agpr_count", 0))
        rows.append((os.path.basename(src), dem[:96], tot - ag if tot >= ag else tot, ag, k.get(".sgpr_count", "?"),
                     k.get(".private_segment_fixed_size", "?"), k.get(".group_segment_fixed_size", "?")))
rows.sort()
path = os.path.join(root, "profiles", (sys.argv[1] if len(sys.argv) > 1 else "r03") + "_kernel_resources.md")
with open(path, "w") as o:
    o.write("# Kernel resources (gfx950 code-object metadata; produced by `tools/kernel_resources.py`, no GPU needed)\n\n")
    o.write("`scratch` = private segment bytes per lane (non-zero = register spills); `LDS` = static only (the scan kernels' rings are dynamic).\n\n")
    o.write("| file | kernel | VGPR | AGPR | SGPR | scratch B | static LDS B |\n|---|---|---|---|---|---|---|\n")
    for r in rows:
        o.write("| " + " | ".join(str(x) for x in r) + " |\n")
print(path, len(rows), "kernels;", sum(1 for r in rows if r[5] not in (0, "0", "?")), "with scratch")
