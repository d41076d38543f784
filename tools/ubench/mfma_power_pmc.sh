# clock and MFMA busy of tools/ubench/mfma_power's variants (PMC pass: --kernel-trace --pmc only)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/mp_pmc
timeout 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d /tmp/mp_pmc -o m -- $R/tools/ubench/mfma_power 0.25 > /tmp/mp_pmc.log 2>&1
cat /tmp/mp_pmc.log | grep TFLOP
python - <<'P'
import csv, glob, collections
f = glob.glob("/tmp/mp_pmc/**/*counter_collection.csv", recursive=True)[0]
kt = glob.glob("/tmp/mp_pmc/**/*kernel_trace.csv", recursive=True)[0]
dur = {r["Dispatch_Id"]: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in csv.DictReader(open(kt))}
acc = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:60]
    d = acc.setdefault(k, collections.defaultdict(float))
    d[r["Counter_Name"]] += float(r["Counter_Value"])
    d["_n_" + r["Counter_Name"]] += 1
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        d["_ns"] += dur.get(r["Dispatch_Id"], 0)
for k, d in acc.items():
    n = d["_n_GRBM_GUI_ACTIVE"]
    gui = d["GRBM_GUI_ACTIVE"] / n
    ns = d["_ns"] / n
    print(f"{k:60s} dispatches {int(n):4d}  mean {ns/1e3:8.1f} us  GRBM_GUI_ACTIVE {gui:.3e} -> clock {gui / ns:.3f} GHz (if the counter is per XCD: x1; summed over 8 XCDs: /8 = {gui / ns / 8:.3f})  MFMA busy / SQ busy {d['SQ_VALU_MFMA_BUSY_CYCLES'] / max(d['SQ_BUSY_CYCLES'], 1):.3f}")
P
