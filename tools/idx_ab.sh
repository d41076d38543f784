#!/bin/bash
# The `index` leg of bench.py (texts -> add_documents -> tokenizer -> encoder -> corpus) with the tokenizer's helper threads capped at
# several values (RMU_TOK_THREADS, a tuning switch), one box: end-to-end rate, the 1000-document pattern, encoder-only, tokenizer-only.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/idx
for t in ${1:-0 128 64 32 16 0}; do
  if [ "$t" = "0" ]; then envs=""; else envs="RMU_TUNING=1 RMU_TOK_THREADS=$t"; fi
  (env $envs timeout 300 python bench.py --rows 1000000 --steps 3 --warmup 1 --legs index --no-cpu-baseline --no-kernel-timing --no-identity-check --index-texts ${TEXTS:-524288} 2>/dev/null | grep "^{" | tail -1) > gpurun_out/idx/ab_$t.json
  python - <<PY
import json
d=json.load(open("gpurun_out/idx/ab_$t.json"))
l=d["secondary"][0]
print("tok_threads=$t (0 = the library's default)", l["value"], l["ms_per_step"], "ref1000", l["reference_pattern_1000_doc_calls"]["chunks_per_sec"], "enc", l["encoder_only"]["chunks_per_sec"], "tok", l["tokenizer_only"]["texts_per_sec"], "ratio", l["end_to_end_over_encoder_only"])
PY
done
