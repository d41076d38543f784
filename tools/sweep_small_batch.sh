#!/bin/bash
# Ladder-geometry sweep for the HBM-bound batch sizes (10M x 384): RMU_SCREEN_RATIO x RMU_SCREEN_MINLVL, merge kernel.
# usage: bash tools/sweep_small_batch.sh <outdir>
export RMU_TUNING=1      # librmu honours its RMU_* switches only with this set
out=${1:-gpurun_out/sweep}; mkdir -p $out
for b in 32 128; do
  for cfg in "0 256" "8 16384" "8 131072" "16 131072" "32 131072" "64 262144"; do
    set -- $cfg
    RMU_SCREEN_RATIO=$1 RMU_SCREEN_MINLVL=$2 timeout 200 python bench.py --batch $b --steps 20 --warmup 3 --legs none --no-cpu-baseline --no-identity-check \
      2> $out/b${b}_r$1_m$2.err | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('b=$b ratio=$1 minlvl=$2', 'qps', d['value'], 'step_ms', d['ms_per_step'], 'kernel_ms', r['kernel_ms'], r['bound'], r['frac'], 'launches', r['launch']['launches'])
" | tee -a $out/summary.txt
  done
done
RMU_MERGE_SELECT=0 timeout 200 python bench.py --batch 32 --steps 20 --legs none --no-cpu-baseline --no-identity-check 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('b=32 merge_select=0 qps', d['value'], 'step_ms', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'])
" | tee -a $out/summary.txt
