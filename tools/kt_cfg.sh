#!/bin/bash
# dev helper: per-kernel totals of a bench run of one config:  tools/kt_cfg.sh C4
set -e -o pipefail
cfg=$1
out=$PWD/gpurun_out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${cfg}_kt -- python bench.py --config $cfg --no-cpu-baseline > $out/${cfg}_kt.json 2> $out/${cfg}_kt.err
python - <<PY
import csv, glob, json
f = glob.glob("$out/${cfg}_kt/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:18]:
    print(r["Name"].split("(")[0][:70], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us", r["Percentage"], "%")
d = json.load(open("$out/${cfg}_kt.json")); print(d["ms_per_step"], d["steps"], d["warmup"])
PY
