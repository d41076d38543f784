#!/bin/bash
# Round profiles: run ON THE GPU BOX through gpurun:  gpurun -- 'bash tools/profile.sh r01'
# Produces gpurun_out/<tag>/*.csv|json; copy the summaries into profiles/ (tracked).
# PMC passes are separate runs with --kernel-trace only (gpurun refuses --pmc together with other trace domains).
# Every rocprofv3 call runs under `timeout`: a PMC pass over the ENCODER benchmark (tools/bench_configs.py embed) aborted inside
# rocprofv3 (signal 6) and then hung in its finalisation until the box's limit -- 15 GPU-minutes lost.  The encoder is profiled
# with --kernel-trace --stats only.
set -u
export RMU_TUNING=1      # librmu honours its RMU_* switches (RMU_SCREEN=0 below) only with this set
TAG=${1:-r05p}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 10 --warmup 2 --legs none --no-cpu-baseline --no-identity-check"
keep() {  # keep only our kernels' rows of a CSV
  f=$(find $OUT/$1 -name "*$2" | head -1)
  if [ -n "$f" ]; then (head -1 $f; grep -E "scan_topk|scan_screen|k_rescore|k_split_rows|k_seed_thr|k_img_err|k_mmr|merge_keys|merge_lists|merge_select|merge_wg|k_gather_flagged|k_ffn3|k_ffn2|k_ffn_fused|k_gemm3|k_gemm|k_attn3|k_qkv_attn_small|k_attention|k_layernorm|k_embed_ln|k_pool|k_cls_head|k_cu_seqlens" $f) | cut -c1-400 > $OUT/$1_$3.csv; fi
}
# 1) headline bench (10M x 384, B=1024) on the default path (fp16 hi/lo screening + exact re-score): stats + PMC passes
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/scan -o scan -- $B > $OUT/scan_bench.json 2> $OUT/scan.err
keep scan kernel_stats.csv stats
timeout 240 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --output-format csv -d $OUT/pmc_a -o a -- $B > $OUT/pmc_a_bench.json 2>> $OUT/scan.err
keep pmc_a counter_collection.csv counters
timeout 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_b -o b -- $B > $OUT/pmc_b_bench.json 2>> $OUT/scan.err
keep pmc_b counter_collection.csv counters
timeout 240 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_c -o c -- $B > $OUT/pmc_c_bench.json 2>> $OUT/scan.err
keep pmc_c counter_collection.csv counters
# 2) the same workload forced onto the exact fp32 scan (RMU_SCREEN=0): stats + MFMA-busy + FETCH_SIZE
export RMU_SCREEN=0
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/exact -o scan -- $B > $OUT/exact_bench.json 2>> $OUT/scan.err
keep exact kernel_stats.csv stats
timeout 240 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --output-format csv -d $OUT/exact_pmc_a -o a -- $B > $OUT/exact_pmc_a_bench.json 2>> $OUT/scan.err
keep exact_pmc_a counter_collection.csv counters
timeout 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/exact_pmc_b -o b -- $B > $OUT/exact_pmc_b_bench.json 2>> $OUT/scan.err
keep exact_pmc_b counter_collection.csv counters
unset RMU_SCREEN
# 3) HBM-bound regime (B=1): default path (fp16 image, screening ladder) and the exact fp32 scan; kernel stats + FETCH_SIZE
B1="python $R/bench.py --batch 1 --steps 10 --warmup 2 --legs none --no-cpu-baseline --no-identity-check"
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/scan_b1 -o scan -- $B1 > $OUT/scan_b1_bench.json 2>> $OUT/scan.err
keep scan_b1 kernel_stats.csv stats
timeout 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_b1 -o b -- $B1 > $OUT/pmc_b1_bench.json 2>> $OUT/scan.err
keep pmc_b1 counter_collection.csv counters
export RMU_SCREEN=0
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/exact_b1 -o scan -- $B1 > $OUT/exact_b1_bench.json 2>> $OUT/scan.err
keep exact_b1 kernel_stats.csv stats
timeout 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/exact_pmc_b1 -o b -- $B1 > $OUT/exact_pmc_b1_bench.json 2>> $OUT/scan.err
keep exact_pmc_b1 counter_collection.csv counters
unset RMU_SCREEN
# 3b) the north-star regime (B=32, screened, nt stream): kernel stats + FETCH_SIZE
B32="python $R/bench.py --batch 32 --steps 10 --warmup 2 --legs none --no-cpu-baseline --no-identity-check"
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/scan_b32 -o scan -- $B32 > $OUT/scan_b32_bench.json 2>> $OUT/scan.err
keep scan_b32 kernel_stats.csv stats
timeout 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_b32 -o b -- $B32 > $OUT/pmc_b32_bench.json 2>> $OUT/scan.err
keep pmc_b32 counter_collection.csv counters
# 3b') batch 16 -- the leg bench.py reports as roofline.north_star (>= 10k queries/sec at >= 0.70 of 8 TB/s): kernel stats + FETCH_SIZE (no b16 record existed)
B16="python $R/bench.py --batch 16 --steps 10 --warmup 2 --legs none --no-cpu-baseline --no-identity-check"
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/scan_b16 -o scan -- $B16 > $OUT/scan_b16_bench.json 2>> $OUT/scan.err
keep scan_b16 kernel_stats.csv stats
timeout 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_b16 -o b -- $B16 > $OUT/pmc_b16_bench.json 2>> $OUT/scan.err
keep pmc_b16 counter_collection.csv counters
# 3c) BASELINE.json configs[1] (1M x 384, batch 1024) and the 8-way shard size (1.25M rows): kernel stats (round 5: VERDICT r4 asked for a c2_stats.csv)
for cfg in "c2 1000000" "shard8 1250000"; do set -- $cfg
  timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$1 -o scan -- python $R/bench.py --rows $2 --steps 10 --warmup 2 --legs none --no-cpu-baseline --no-identity-check > $OUT/$1_bench.json 2>> $OUT/scan.err
  keep $1 kernel_stats.csv stats
done
# 4) encoder (config 3) kernel stats (kernel-trace only: a PMC pass over the encoder hung rocprofv3 in round 1)
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/embed -o e -- python $R/bench.py --legs embed --rows 100000 --steps 2 --no-cpu-baseline --no-identity-check --no-kernel-timing > $OUT/embed_bench.json 2>> $OUT/scan.err
keep embed kernel_stats.csv stats
# 5) C5's dense part: 64 queries x 1M rows, top-100 (the deep-k threshold ladder): kernel stats
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/top100 -o t -- python $R/tools/topk_probe.py 1000000:64:100 > $OUT/top100_probe.txt 2>> $OUT/scan.err
keep top100 kernel_stats.csv stats
# 6) encoder PMC (the four dominant kernels only; a pass over every kernel hung rocprofv3 in round 1)
timeout -k 5 150 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-include-regex "k_ffn3|k_gemm3|k_attn3|k_gemm" --output-format csv -d $OUT/enc_pmc -o a -- python $R/tools/enc_smoke.py 2048 > /dev/null 2>> $OUT/scan.err
keep enc_pmc counter_collection.csv counters
# 7) encoder HBM traffic (round 4): FETCH_SIZE and WRITE_SIZE, separate passes, over ONE forward of exactly the workloads bench.py's
# embed (C3: 8192 chunks) and rerank (C5: 6400 pairs) legs time -- every encoder kernel of the forward (k_* only)
ENCK="k_ffn3|k_gemm3|k_attn3|k_gemm|k_embed_ln|k_pool|k_layernorm|k_cls_head|k_cu_seqlens"
for wl in "embed 8192" "rerank 6400"; do set -- $wl; name=$1; n=$2; extra=""; [ $name = rerank ] && extra="0 pair";
  ENC_DEVICE_IDS=1 timeout -k 5 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE --kernel-include-regex "$ENCK" --output-format csv -d $OUT/enc_${name}_f -o a -- python $R/tools/enc_smoke.py $n $extra > /dev/null 2>> $OUT/scan.err
  keep enc_${name}_f counter_collection.csv counters
  ENC_DEVICE_IDS=1 timeout -k 5 150 rocprofv3 --kernel-trace --pmc WRITE_SIZE --kernel-include-regex "$ENCK" --output-format csv -d $OUT/enc_${name}_w -o a -- python $R/tools/enc_smoke.py $n $extra > /dev/null 2>> $OUT/scan.err
  keep enc_${name}_w counter_collection.csv counters
done
for d in scan_b16 pmc_b16 c2 shard8 enc_embed_f enc_embed_w enc_rerank_f enc_rerank_w top100 enc_pmc scan pmc_a pmc_b pmc_c exact exact_pmc_a exact_pmc_b scan_b1 pmc_b1 exact_b1 exact_pmc_b1 scan_b32 pmc_b32 embed; do rm -rf $OUT/$d; done
ls -la $OUT
