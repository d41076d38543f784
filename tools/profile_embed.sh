#!/bin/bash
# Encoder-only refresh of the round profiles (run ON THE GPU BOX): kernel-trace statistics of the C3 embed leg.
set -u
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/embed -o e -- python $R/bench.py --legs embed --rows 100000 --steps 2 --no-cpu-baseline --no-identity-check --no-kernel-timing > $OUT/embed_bench.json 2>> $OUT/scan.err
f=$(find $OUT/embed -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then (head -1 $f; grep -E "k_ffn_fused|k_gemm3|k_gemm|k_attention|k_layernorm|k_embed_ln|k_meanpool|k_cls_head" $f) | cut -c1-400 > $OUT/embed_stats.csv; fi
rm -rf $OUT/embed
cat $OUT/embed_stats.csv | cut -c1-160
