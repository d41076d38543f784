set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; export TMPDIR=/tmp
for b in 16 32 1024; do
  rm -rf /tmp/st$b
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/st$b -o s -- python $R/tools/search_trace.py run $b ${1:-10000000} > /dev/null 2>&1)
  echo "== batch $b"; python tools/search_trace.py show /tmp/st$b | grep -E "k_rescore|span"
done
