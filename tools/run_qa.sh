set -x
timeout 900 python -m pytest tests/test_encoder_gpu.py -m gpu -x -q -k "bit_identical" 2>&1 | tail -3
for v in 0 1; do
  RMU_TUNING=1 RMU_QA=$v timeout 600 python bench.py --rows 200000 --legs embed --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['secondary'][0]; print('QA=$v', s['value'], s['ms_per_step'], s['roofline']['frac'])"
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_qa -o qa -- python $GRAFT_REPO_ROOT/bench.py --rows 200000 --legs embed --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(ls gpurun_out/prof_qa/*kernel_stats.csv gpurun_out/prof_qa/*/*kernel_stats.csv 2>/dev/null | head -1)
echo $f; head -12 "$f" | cut -c1-200
