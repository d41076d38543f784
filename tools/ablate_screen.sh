#!/bin/bash
# Timing ablations of scan_screen_kernel.  Needs the debug instantiations: build with
#   python -m ragmeup_amd.build --debug-kernels      (then rebuild the product library: python -m ragmeup_amd.build --force)
# and force the 4-wave form the table was measured on: RMU_SCREEN_W8=0.  Run ON THE GPU BOX:  gpurun -- 'bash tools/ablate_screen.sh'
# RMU_SCREEN_EXP bits: 1 = no corpus LDS-DMA, 2 = no LDS fragment reads, 8 = no filter compares (results are wrong by design);
# RMU_SCREEN_NOFILTER=1 skips the candidate appends; RMU_SCAN_EXP=7 selects the build with cycle counters (clock64 per wave).
# Prints, for the largest row range of the 10M x 1024 ladder: wall time of the launch (rocprofv3), cycles per wave, clock.
export RMU_TUNING=1      # librmu honours its RMU_* switches only with this set
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
export RMU_SCREEN_NOFILTER=1 RMU_SCAN_EXP=7
for e in 0 8 9 10 11; do
  export RMU_SCREEN_EXP=$e
  rocprofv3 --kernel-trace --output-format csv -d /tmp/abl_$e -o a -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> /tmp/abl_$e.err
  f=$(find /tmp/abl_$e -name "*kernel_trace.csv" | head -1)
  python - "$f" $e /tmp/abl_$e.err <<'PY'
import csv, re, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'scan_screen' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
dur = (int(rows[-1]['End_Timestamp']) - int(rows[-1]['Start_Timestamp'])) / 1e3
last = [l for l in open(sys.argv[3]) if 'rmu dbg range' in l][-1]
m = re.search(r'clk_bar=(\d+) clk_all=(\d+)', last)
bar, allc = int(m.group(1)) / 1024, int(m.group(2)) / 1024
names = {0: 'production loop (appends off)', 8: 'no filter compares', 9: 'no DMA, no compares', 10: 'no LDS reads, no compares', 11: 'MFMA + loop only'}
print(f"EXP {sys.argv[2]:>2} {names[int(sys.argv[2])]:32s} launch {dur:7.0f} us  cycles/wave {allc:9.0f}  at barrier {bar:8.0f}  clock {allc/dur/1e3:.2f} GHz")
PY
done
