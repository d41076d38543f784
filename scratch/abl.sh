cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() {
  python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline $2 > /tmp/tr_$1.json 2>/tmp/tr_$1.err
  python - $1 /tmp/tr_$1.json <<'PY'
import sys,json
try:
    b=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(sys.argv[1], 'step_ms', b['ms_per_step'], 'kernel_ms', b['roofline']['kernel_ms'], 'launches', b['roofline']['launch']['launches'], 'qps', b['value'], 'rerun', b['roofline'].get('rerun_queries'))
except Exception as e:
    print(sys.argv[1], 'FAILED', e); print(open(sys.argv[2].replace('.json','.err')).read()[-600:])
PY
}
run base
RMU_SCREEN_NOFILTER=1 run nofilter
run base1m "--rows 1000000"
run b512 "--batch 512"
run b2048 "--batch 2048"
