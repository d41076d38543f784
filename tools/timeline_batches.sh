set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; export TMPDIR=/tmp
for b in $1; do
  rm -rf /tmp/st$b
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/st$b -o s -- python $R/tools/search_trace.py run $b ${2:-10000000} > /dev/null 2>&1)
  echo "== batch $b rows ${2:-10000000}"; python tools/search_trace.py show /tmp/st$b | cut -c1-110
done
