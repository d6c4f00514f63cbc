#!/bin/bash
# dev helper (run on the GPU box from the repo root): the evidence set of profiles/ for one state of the code
#   tools/final_profile.sh <tag>      -> gpurun_out/<tag>_*
set -e -o pipefail
tag=$1
out=$PWD/gpurun_out
mkdir -p $out
export TMPDIR=/tmp
python bench.py --steps 10 --warmup 2 > $out/${tag}_bench_c2.json 2> $out/${tag}_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_kt -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-host-path > $out/${tag}_bench_c2_under_rocprof.json 2> $out/${tag}_kt.err
cp $out/${tag}_kt/*/*kernel_stats.csv $out/${tag}_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/${tag}_pf -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-path > $out/${tag}_pf.json 2> $out/${tag}_pf.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/${tag}_pw -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-path > $out/${tag}_pw.json 2> $out/${tag}_pw.err
python tools/pmc_traffic.py $out/${tag}_pf/*/*counter_collection.csv $out/${tag}_pw/*/*counter_collection.csv $out/${tag}_pf.json $out/${tag}_pmc_traffic.json > /dev/null
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_LDS --output-format csv -d $out/${tag}_sq -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-path > $out/${tag}_sq.json 2> $out/${tag}_sq.err
python - <<PY > $out/${tag}_sq_counters.txt
import csv, glob, collections
f = glob.glob("$out/${tag}_sq/*/*counter_collection.csv")[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_INSTS_VALU": n[k] += 1
print("per-launch averages of rocprofv3 --pmc SQ counters (python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-path)")
for k in sorted(acc, key=lambda k: -acc[k]["SQ_BUSY_CYCLES"])[:12]:
    print(k, "launches", n[k], {c: round(v / max(n[k], 1)) for c, v in sorted(acc[k].items())})
PY
tail -c 600 $out/${tag}_bench_c2.json
