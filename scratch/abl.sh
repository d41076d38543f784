cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() {
  python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline $2 > /tmp/tr_$1.json 2>/tmp/tr_$1.err
  python - $1 /tmp/tr_$1.json <<'PY'
import sys,json
try:
    b=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(sys.argv[1], 'step_ms', b['ms_per_step'], 'kernel_ms', b['roofline']['kernel_ms'], 'launches', b['roofline']['launch']['launches'], 'qps', b['value'], 'path', b['roofline']['path'], 'ident', b.get('identical_to_exact_f32_scan'))
except Exception as e:
    print(sys.argv[1], 'FAILED', e); print(open(sys.argv[2].replace('.json','.err')).read()[-600:])
PY
}
for b in 1 32; do
  run r8_b$b "--batch $b"
  RMU_SCREEN_RATIO=16 run r16_b$b "--batch $b"
  RMU_SCREEN_RATIO=64 run r64_b$b "--batch $b"
  RMU_SCREEN_RATIO=1 run r1_b$b "--batch $b"
done
run b1_1m "--batch 1 --rows 1000000"
RMU_SCREEN=0 run b1_1m_exact "--batch 1 --rows 1000000"
