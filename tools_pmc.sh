#!/bin/bash
# PMC passes for the scan kernel (separate passes; --pmc only with --kernel-trace)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_r01
mkdir -p $OUT
ARGS="--steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ_[A-Z_0-9]+|GRBM_[A-Z_0-9]+|TCC_[A-Za-z_0-9]+|FETCH_SIZE|WRITE_SIZE|TCP_[A-Za-z_0-9]+|LDSBankConflict|MfmaUtil|VALUBusy|OccupancyPercent)\b" | sort -u > $OUT/counters_available.txt
wc -l $OUT/counters_available.txt
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --output-format csv -d $OUT/a -o a -- python $R/bench.py $ARGS > $OUT/a.json 2> $OUT/a.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/b -o b -- python $R/bench.py $ARGS > $OUT/b.json 2> $OUT/b.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/c -o c -- python $R/bench.py $ARGS > $OUT/c.json 2> $OUT/c.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD --output-format csv -d $OUT/d -o d -- python $R/bench.py $ARGS > $OUT/d.json 2> $OUT/d.err
ls -R $OUT | head -30
# keep only the counter CSVs (small) -- strip the torch kernels
for p in a b c d; do
  f=$(find $OUT/$p -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then (head -1 $f; grep scan_topk $f) > $OUT/${p}_scan_counters.csv; fi
  rm -rf $OUT/$p
done
ls -la $OUT
