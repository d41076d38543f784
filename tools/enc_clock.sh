#!/bin/bash
# clock and MFMA-busy of the encoder's kernels, ONE PMC pass over a single 8192-chunk forward (run on the GPU box): bash tools/enc_clock.sh
# (timeout-wrapped: an encoder PMC pass once hung inside rocprofv3's finalisation -- tools/profile.sh)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=${1:-/tmp/enc_clk}
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-include-regex "k_ffn3|k_gemm3|k_attn3|k_gemm<|k_qa" \
    --output-format csv -d $OUT -o a -- python $R/tools/enc_smoke.py ${ENC_N:-8192} 0 > $OUT/log.txt 2>&1
f=$(find $OUT -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
by = collections.defaultdict(dict)
for r in rows:
    k = r["Dispatch_Id"]
    by[k][r["Counter_Name"]] = float(r["Counter_Value"])
    by[k]["dur"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    by[k]["name"] = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:28]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for d in by.values():
    a = agg[d["name"]]
    a["n"] += 1
    for k, v in d.items():
        if k != "name": a[k] += v
for name, a in sorted(agg.items(), key=lambda kv: -kv[1]["dur"]):
    gui = a["GRBM_GUI_ACTIVE"]
    print(f'{name:28s} launches {a["n"]:3.0f} avg {a["dur"] / a["n"]:8.1f} us  clock {gui / 8 / a["dur"] / 1e3:5.2f} GHz  mfma_busy {a["SQ_VALU_MFMA_BUSY_CYCLES"] / max(gui, 1) / 128:6.3f}  '
          f'valu/wave-cycle {a["SQ_INSTS_VALU"] / max(a["SQ_WAVE_CYCLES"], 1):.4f}  wait_inst {a["SQ_WAIT_INST_ANY"] / max(a["SQ_WAVE_CYCLES"], 1):.3f}  active_inst {a["SQ_ACTIVE_INST_ANY"] / max(a["SQ_WAVE_CYCLES"], 1):.3f}')
PY
