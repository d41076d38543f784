#!/bin/bash
# per-kernel durations of one small-batch search step over 10M rows (the north-star regime): bash tools/small_batch_stats.sh 16 32
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for b in "$@"; do
  rm -rf /tmp/sb_$b
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sb_$b -o s -- python $R/bench.py --batch $b --steps 20 --warmup 3 --legs none --no-cpu-baseline --no-identity-check > /tmp/sb_$b.json 2>/tmp/sb_$b.err
  python - <<PY
import csv,glob,json
d=json.loads(open("/tmp/sb_$b.json").read().strip().splitlines()[-1])
print("== batch $b: step %.4f ms, %.1f q/s, kernel_ms %s, frac %s" % (d["ms_per_step"], d["value"], d["roofline"]["kernel_ms"], d["roofline"]["frac"]))
f=glob.glob("/tmp/sb_$b/**/*kernel_stats.csv",recursive=True)
rows=[r for r in csv.DictReader(open(f[0]))]
steps=33.0   # 20 timed + 3 warm-up + 10 roofline passes (bench.py) -- printed per call anyway
for r in rows[:14]:
    print("  %-70s calls %5s avg_us %8.2f total_ms %8.3f" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
done
