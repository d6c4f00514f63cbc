#!/bin/bash
# dev helper: per-kernel averages of a C3 bench run
set -e -o pipefail
out=$PWD/gpurun_out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/c3_kt -- python bench.py --config C3 --no-cpu-baseline > $out/c3_kt.json 2> $out/c3_kt.err
python - <<PY
import csv, glob
f = glob.glob("$out/c3_kt/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print(r["Name"].split("(")[0][:80], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us")
PY
