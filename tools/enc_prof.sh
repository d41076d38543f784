#!/bin/bash
# Per-kernel durations of the encoder for a list of RMU_GEMM2 masks (run ON THE GPU BOX):  bash tools/enc_prof.sh 7 3 0
export RMU_TUNING=1      # librmu honours its RMU_* switches only with this set
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for g in "$@"; do
  RMU_GEMM3=$g TAG=g2_$g timeout 200 env ${EXTRA_ENV:-X=0} rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p$g -o x -- python $R/tools/enc_smoke.py ${ENC_N:-8192} 5 > /tmp/log$g 2>&1
  grep RATE /tmp/log$g
  python - <<PY
import csv,glob
f=glob.glob("/tmp/p$g/**/*kernel_stats.csv",recursive=True)
if not f: print(open("/tmp/log$g").read()[-1500:])
else:
    for r in list(csv.DictReader(open(f[0])))[:10]:
        print("  %-72s %5s avg_us %8.1f pct %s" % (r["Name"][:72], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
done
