#!/bin/bash
# A/B of encoder kernel variants ON THE GPU BOX: per-kernel durations (rocprofv3 --kernel-trace --stats) and the encode rate of
# tools/enc_smoke.py for each "name:VAR=val,VAR=val" spec.   bash tools/enc_ab.sh default: ffn2:RMU_FFN_V=2 rowmajor:RMU_H_TILED=0,RMU_CTX_TILED=0
export RMU_TUNING=1      # librmu honours its RMU_* switches only with this set
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do
  name=${spec%%:*}; envs=${spec#*:}; envs=${envs//,/ }
  echo "== $name  [$envs]"
  TAG=$name timeout 240 env $envs rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$name -o x -- python $R/tools/enc_smoke.py ${ENC_N:-8192} 5 > /tmp/log_$name 2>&1
  grep -E "RATE|OK " /tmp/log_$name
  python - <<PY
import csv,glob
f=glob.glob("/tmp/p_$name/**/*kernel_stats.csv",recursive=True)
if not f: print(open("/tmp/log_$name").read()[-1500:])
else:
    for r in list(csv.DictReader(open(f[0])))[:8]:
        print("  %-60s %5s avg_us %8.1f pct %s" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
done
