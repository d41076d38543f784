#!/bin/bash
# clock and MFMA-busy of the screening kernel forms, one PMC pass each (run on the GPU box): bash tools/pmc_clock.sh <outdir>
# per form, the longest launches: duration, GRBM_GUI_ACTIVE (summed over 8 XCDs) / 8 / duration = the clock the kernel got, MFMA busy cycles / (GUI_ACTIVE / 8 x 1024 SIMDs)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/${1:-gpurun_out/r04clk}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {  # name, env...
  name=$1; shift
  env RMU_TUNING=1 "$@" timeout 240 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-include-regex "scan_screen" \
      --output-format csv -d $OUT/$name -o a -- python $R/tools/pace_probe.py --env RMU_X --pace 0 --steps 3 > $OUT/$name.txt 2>&1
  f=$(find $OUT/$name -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$name" <<'PY' | tee -a $OUT/summary.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
by = collections.defaultdict(dict)
for r in rows:
    k = (r["Dispatch_Id"])
    by[k][r["Counter_Name"]] = float(r["Counter_Value"])
    by[k]["dur"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    by[k]["name"] = r["Kernel_Name"][:60]
big = sorted(by.values(), key=lambda d: -d["dur"])[:6]
for d in big:
    gui = d.get("GRBM_GUI_ACTIVE", 0)
    print(f'{sys.argv[2]:8s} dur {d["dur"]:8.1f} us  GUI_ACTIVE {gui:12.0f}  clock {gui / 8 / d["dur"] / 1e3:5.2f} GHz  mfma_busy {d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(gui, 1) / 128:6.3f} of SIMD-cycles  sq_busy {d.get("SQ_BUSY_CYCLES", 0) / max(gui, 1):8.2f} valu_insts {d.get("SQ_INSTS_VALU", 0):.3g} wave_cyc {d.get("SQ_WAVE_CYCLES", 0):.3g} wait_inst {d.get("SQ_WAIT_INST_ANY", 0):.3g} active_inst {d.get("SQ_ACTIVE_INST_ANY", 0):.3g}')
PY
  rm -rf $OUT/$name
}
# forms: round3 | lean | lean2 | lean3 (product default) -- product library; ks | kpp | g4 -- debug builds only (python -m ragmeup_amd.build --debug-kernels)
FORMS=${2:-"round3 lean lean2 lean3"}
for f in $FORMS; do
  case $f in
    round3|old) run round3 RMU_SCREEN_LEAN=0;;
    lean)  run lean RMU_SCREEN_LEAN=1;;
    lean2) run lean2 RMU_SCREEN_LEAN=2;;
    lean3) run lean3 RMU_SCREEN_LEAN=3;;
    ks)  run ks RMU_SCREEN_KS=1 RMU_SCREEN_KPP=0;;
    kpp) run kpp RMU_SCREEN_KS=1 RMU_SCREEN_KPP=1;;
    g4)  run g4 RMU_SCREEN_G4=1;;
    *) echo "unknown form $f";;
  esac
done
