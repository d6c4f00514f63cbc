#!/bin/bash
# dev helper (GPU box): same-box A/B of two builds of libpfbhip.so -- tools/ab/libpfbhip_{old,new}.so, alternating, C2 bench
set -e
out=gpurun_out/ab
mkdir -p $out
for rep in 1 2; do
for v in old new; do
  cp tools/ab/libpfbhip_$v.so pfb-imaging_amd/libpfbhip.so
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-host-path > $out/${v}_$rep.json 2>/dev/null
  PFBHIP_WMODE2=0 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-path > $out/${v}_wm1_$rep.json 2>/dev/null
  python - <<PY
import json
for f in ("${v}_$rep", "${v}_wm1_$rep"):
    d = json.load(open("$out/%s.json" % f)); s = d["roofline"]["stage_ms_per_step"]
    print(f, round(d["ms_per_step"], 3), "grid", s["grid"], "degrid", s["degrid"])
PY
done
done
