# full GPU validation at HEAD: suite, smoke, driver-equivalent bench, small-batch timelines (bash tools/validate.sh <tag>)
set -u
TAG=${1:-val}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 1200 python -m pytest tests -m gpu -q -x --timeout=300 2>&1 | grep -E "passed|failed|error|Timeout" | tail -5 | tee gpurun_out/${TAG}_gpu_tests.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python bench.py > gpurun_out/${TAG}_bench_line.json 2> gpurun_out/${TAG}_bench_err.txt; wc -c gpurun_out/${TAG}_bench_line.json
cp bench_secondary.json gpurun_out/${TAG}_bench_full.json 2>/dev/null
export TMPDIR=/tmp
for b in 16 32 1024; do
  rm -rf /tmp/st$b
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/st$b -o s -- python $R/tools/search_trace.py run $b > /dev/null 2>&1)
  echo "== batch $b over 10M rows: kernels of the last search"
  python tools/search_trace.py show /tmp/st$b
done > gpurun_out/${TAG}_search_timeline.txt 2>&1
