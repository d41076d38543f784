cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() {
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abl_$1 -o a -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > /tmp/abl_$1.json 2>/dev/null
  f=$(find /tmp/abl_$1 -name "*kernel_stats.csv" | head -1)
  echo "$1: $(grep scan_screen $f | sed "s/.*ScanLaunch)\",//" | cut -d, -f1-3)  step_ms=$(python -c "import json;print(json.loads(open('/tmp/abl_$1.json').read().strip().splitlines()[-1])['ms_per_step'])")"
}
run base
RMU_SCREEN_PRE=0 run nopre
RMU_SCREEN_PRE=6 run pre6
RMU_SCREEN_PRE=24 run pre24
RMU_SCREEN_NOFILTER=1 run nofilter
RMU_SCREEN_NOFILTER=1 RMU_SCREEN_EXP=1 run nodma
RMU_SCREEN_NOFILTER=1 RMU_SCREEN_EXP=2 run nolds
RMU_SCREEN_NOFILTER=1 RMU_SCREEN_EXP=3 run mfmaonly
