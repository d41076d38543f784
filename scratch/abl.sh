cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() {
  python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline $2 > /tmp/tr_$1.json 2>/dev/null
  python - $1 /tmp/tr_$1.json <<'PY'
import sys,json
b=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(sys.argv[1], 'step_ms', b['ms_per_step'], 'kernel_ms', b['roofline']['kernel_ms'], 'launches', b['roofline']['launch']['launches'], 'qps', b['value'])
PY
}
run base
RMU_SCREEN_RATIO=2 run r2
RMU_SCREEN_RATIO=3 run r3
RMU_SCREEN_MINLVL=8192 run r4m8k
RMU_SCREEN_MINLVL=8192 RMU_SCREEN_RATIO=2 run r2m8k
RMU_SCREEN_MINLVL=8192 RMU_SCREEN_RATIO=3 run r3m8k
RMU_SCREEN_MINLVL=2048 RMU_SCREEN_RATIO=3 run r3m2k
RMU_SCREEN_NOFILTER=1 run nofilter
RMU_SCAN_EXP=7 python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline 2>&1 | grep "rmu dbg" | tail -6
