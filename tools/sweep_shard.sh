#!/bin/bash
# Ladder shape sweep (run ON THE GPU BOX):  ROWS=1250000 bash tools/sweep_shard.sh   (8-way shard size; default) or ROWS=10000000.
# Prints ms per step and qps per setting.
export RMU_TUNING=1      # librmu honours its RMU_* switches only with this set
B="python bench.py --rows ${ROWS:-1250000} --steps ${STEPS:-40} --warmup 5 --legs none --no-cpu-baseline --no-identity-check --no-kernel-timing"
run() { printf "%-44s " "$*"; env "$@" $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"; }
run RMU_X=0
run RMU_SCREEN_RATIO=2
run RMU_SCREEN_RATIO=4
run RMU_SCREEN_RATIO=5
run RMU_SCREEN_RATIO=3 RMU_SCREEN_MINLVL=64
run RMU_SCREEN_RATIO=3 RMU_SCREEN_MINLVL=1024
run RMU_X=0
