#!/bin/bash
# Round profiles: run ON THE GPU BOX through gpurun:  gpurun -- 'bash tools/profile.sh r01'
# Produces gpurun_out/<tag>/*.csv|json; copy the summaries into profiles/ (tracked).
# PMC passes are separate runs with --kernel-trace only (gpurun refuses --pmc together with other trace domains).
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline"
keep() {  # keep only our kernels' rows of a CSV
  f=$(find $OUT/$1 -name "*$2" | head -1)
  if [ -n "$f" ]; then (head -1 $f; grep -E "scan_topk|merge_keys|k_gemm|k_attention|k_layernorm|k_embed_ln|k_meanpool|k_cls_head" $f) | cut -c1-400 > $OUT/$1_$3.csv; fi
}
# 1) kernel trace + stats of the headline bench (10M x 384, B=1024)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/scan -o scan -- $B > $OUT/scan_bench.json 2> $OUT/scan.err
keep scan kernel_stats.csv stats
# 2) PMC passes for the scan kernel
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --output-format csv -d $OUT/pmc_a -o a -- $B > $OUT/pmc_a_bench.json 2>> $OUT/scan.err
keep pmc_a counter_collection.csv counters
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_b -o b -- $B > $OUT/pmc_b_bench.json 2>> $OUT/scan.err
keep pmc_b counter_collection.csv counters
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_c -o c -- $B > $OUT/pmc_c_bench.json 2>> $OUT/scan.err
keep pmc_c counter_collection.csv counters
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD --output-format csv -d $OUT/pmc_d -o d -- $B > $OUT/pmc_d_bench.json 2>> $OUT/scan.err
keep pmc_d counter_collection.csv counters
# 3) HBM-bound regime (B=1): kernel stats + FETCH_SIZE
B1="python $R/bench.py --batch 1 --steps 10 --warmup 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/scan_b1 -o scan -- $B1 > $OUT/scan_b1_bench.json 2>> $OUT/scan.err
keep scan_b1 kernel_stats.csv stats
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_b1 -o b -- $B1 > $OUT/pmc_b1_bench.json 2>> $OUT/scan.err
keep pmc_b1 counter_collection.csv counters
# 4) encoder (config 3) kernel stats
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/embed -o e -- python $R/tools/bench_configs.py embed --chunks 32768 --batch 8192 > $OUT/embed_bench.json 2>> $OUT/scan.err
keep embed kernel_stats.csv stats
for d in scan pmc_a pmc_b pmc_c pmc_d scan_b1 pmc_b1 embed; do rm -rf $OUT/$d; done
ls -la $OUT
