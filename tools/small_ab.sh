#!/bin/bash
# bash tools/small_ab.sh  -> one line per switch setting (fresh process each: the switches are read once)
export RMU_TUNING=1
R=${GRAFT_REPO_ROOT:-$(pwd)}
for spec in "default:" "fused_from_first_token:RMU_QKV_ATTN_MIN=0" "no_qkv_attn:RMU_QKV_ATTN_TOKENS=0" "no_small_fuse:RMU_SMALL_FUSE=0" "round4:RMU_QKV_ATTN_TOKENS=0,RMU_SMALL_FUSE=0" "qkv_attn_to_4096:RMU_QKV_ATTN_TOKENS=4096" "default2:"; do
  name=${spec%%:*}; envs=${spec#*:}; envs=${envs//,/ }
  timeout 200 env $envs python $R/tools/small_ab.py $name 2>&1 | grep -E "^AB|Error|error" | head -5
done
