#!/bin/bash
# dev helper (GPU box, repo root): per-kernel average durations of a short C2 bench run -> gpurun_out/<tag>_kt.txt
set -e -o pipefail
tag=$1
out=$PWD/gpurun_out
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_kt -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-path > $out/${tag}_kt.json 2> $out/${tag}_kt.err
python - <<PY > $out/${tag}_kt.txt
import csv, glob
f = glob.glob("$out/${tag}_kt/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print(r["Name"].split("(")[0][:90], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us")
PY
cat $out/${tag}_kt.txt
