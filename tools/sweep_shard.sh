#!/bin/bash
# Ladder shape for the 8-way shard size (1.25M rows, batch 1024): run ON THE GPU BOX.  Prints ms per step per setting.
B="python bench.py --rows ${ROWS:-1250000} --steps 40 --warmup 5 --legs none --no-cpu-baseline --no-identity-check --no-kernel-timing"
run() { printf "%-40s " "$*"; env "$@" $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"; }
run RMU_X=0
run RMU_SCREEN_RATIO=4
run RMU_SCREEN_RATIO=6
run RMU_SCREEN_RATIO=8
run RMU_SCREEN_RATIO=16
run RMU_SCREEN_RATIO=3 RMU_SCREEN_MINLVL=4096
run RMU_SCREEN_RATIO=6 RMU_SCREEN_MINLVL=4096
run RMU_SCREEN_RATIO=8 RMU_SCREEN_MINLVL=32768
run RMU_X=0
