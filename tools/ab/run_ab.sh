#!/bin/bash
# dev helper (GPU box): same-box A/B of two builds of libpfbhip.so -- tools/ab/libpfbhip_{old,new}.so, alternating, C2 bench
#   bash tools/ab/run_ab.sh [reps] [wm1]      (wm1: also the polynomial-plane plan, PFBHIP_WMODE2=0)
set -e
reps=${1:-2}
out=gpurun_out/ab
mkdir -p $out
for rep in $(seq 1 $reps); do
for v in old new; do
  cp tools/ab/libpfbhip_$v.so pfb-imaging_amd/libpfbhip.so
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-host-path > $out/${v}_$rep.json 2>/dev/null
  files="${v}_$rep"
  if [ "$2" = "wm1" ]; then
    PFBHIP_WMODE2=0 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-path > $out/${v}_wm1_$rep.json 2>/dev/null
    files="$files ${v}_wm1_$rep"
  fi
  python - <<PY
import json
for f in "$files".split():
    d = json.load(open("$out/%s.json" % f)); s = d["roofline"]["stage_ms_per_step"]
    print(f, round(d["ms_per_step"], 3), "grid", s["grid"], "degrid", s["degrid"])
PY
done
done
