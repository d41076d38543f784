set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 600 python -m pytest tests/test_search_gpu.py -x -q --timeout=120 2>&1 | grep -E "passed|failed|error|Timeout" | tail -4
for b in 16 32 1024; do
  rm -rf /tmp/st$b
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/st$b -o s -- python $R/tools/search_trace.py run $b > /dev/null 2>&1)
  echo "== batch $b over 10M rows: kernels of the last search"
  python tools/search_trace.py show /tmp/st$b
done > gpurun_out/s2_search_timeline.txt 2>&1
cat gpurun_out/s2_search_timeline.txt | grep -E "==|span|lean3" | head -40
timeout 400 python bench.py --legs b16,b32,b128,b1,c2,emu8,exact --no-cpu-baseline > gpurun_out/s2c_bench_line.json 2> gpurun_out/s2c_bench_err.txt
python -c "
import json
d=json.load(open('gpurun_out/s2c_bench_line.json'))
print(d['value'], d['roofline']['north_star'])
for l in d['secondary']: print(l['id'], l['value'], l['ms_per_step'], l['roofline']['frac'])
"
