#!/bin/bash
# dev helper (GPU box, repo root): bench lines of the other BASELINE configs on one GPU -> gpurun_out/<tag>_bench_<cfg>.json
set -e -o pipefail
tag=$1
out=$PWD/gpurun_out
mkdir -p $out
for cfg in C1 C3 C4 C5; do
  python bench.py --config $cfg --no-cpu-baseline > $out/${tag}_bench_${cfg}.json 2> $out/${tag}_bench_${cfg}.err
  python - <<PY
import json
d = json.load(open("$out/${tag}_bench_${cfg}.json"))
print("$cfg", d["metric"][:60], round(d["value"], 1), d["unit"], round(d["ms_per_step"], 3), "ms/step")
PY
done
