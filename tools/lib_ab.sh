# same-box A/B of two builds of librmu (RMU_LIB needs RMU_TUNING=1): bash tools/lib_ab.sh <lib A> <lib B> [legs]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
LEGS=${3:-b16,b32,c2,emu8}
for rep in 1 2; do
for L in "$1" "$2"; do
  RMU_TUNING=1 RMU_LIB=$R/$L timeout 300 python $R/bench.py --legs $LEGS --no-cpu-baseline --no-identity-check > /tmp/ab.json 2> /tmp/ab.err
  python - "$L" <<'P'
import json, sys
d = json.load(open('/tmp/ab.json'))
print(sys.argv[1].split('/')[-1], "headline %.1f q/s %.4f ms" % (d['value'], d['ms_per_step']), " ".join("%s %.4f" % (l['id'], l['ms_per_step']) for l in d['secondary']), flush=True)
P
done
done
