#!/bin/bash
# dev helper (GPU box): C2 bench line under a list of environment variants, one summary line each
#   tools/bench_variants.sh <outdir> "VAR=val VAR2=val" "VAR=val" ...      ("-" = no variables)
out=$1; shift
mkdir -p $out
i=0
for v in "$@"; do
  i=$((i+1))
  if [ "$v" = "-" ]; then envs=""; else envs="$v"; fi
  env $envs python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-host-path > $out/v$i.json 2> $out/v$i.err || { echo "variant '$v' FAILED"; tail -3 $out/v$i.err; continue; }
  python - "$out/v$i.json" "$v" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
st = d["roofline"]["stage_ms_per_step"]
print(f"{sys.argv[2]:40s} ms/step {d['ms_per_step']:7.3f}  " + "  ".join(f"{k} {v:.3f}" for k, v in st.items() if v))
PY
done
