#!/bin/bash
# dev helper (GPU box, repo root): rocprofv3 kernel statistics beyond the C2 headline -- the PSF-approximate Hessian, the
# wavelet dictionary, the device-resident primal-dual iteration (C4) and the wide-field config C5 -- plus the bench lines of
# C1 / C3 / C4 / C5 (with roofline and cpu_baseline).      tools/profile_others.sh <tag>   -> gpurun_out/<tag>_*
# (the interpreter comes directly after `--`: no env / bash -c hop under rocprofv3)
set -o pipefail
tag=$1
out=$PWD/gpurun_out
mkdir -p $out
export TMPDIR=/tmp
kstats() {  # kstats <name> <program args...>
  name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_${name}_kt -- python "$@" > $out/${tag}_${name}_under_rocprof.json 2> $out/${tag}_${name}_kt.err \
    && cp $out/${tag}_${name}_kt/*/*kernel_stats.csv $out/${tag}_${name}_kernel_stats.csv && echo "$name: ok" || echo "$name: FAILED"
}
kstats psfconv tools/bench_psfconv.py --steps 10 --no-cpu
kstats psi tools/bench_psi.py
kstats pd bench.py --config C4 --steps 5 --warmup 1 --no-cpu-baseline
kstats C5 bench.py --config C5 --steps 2 --warmup 1 --no-cpu-baseline
for cfg in C1 C3 C4 C5; do
  python bench.py --config $cfg > $out/${tag}_bench_${cfg}.json 2> $out/${tag}_bench_${cfg}.err
  python - <<PY
import json
try:
    d = json.load(open("$out/${tag}_bench_${cfg}.json"))
    print("$cfg", d["metric"][:60], round(d["value"], 1), d["unit"], round(d["ms_per_step"], 3), "ms/step", "roofline" in d, "cpu_baseline" in d)
except Exception as e:
    print("$cfg FAILED", e)
PY
done
