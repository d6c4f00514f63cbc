#!/bin/bash
# dev helper (GPU box): C5 bench line under a list of environment variants, one summary line each
#   tools/c5_variants.sh <outdir> "VAR=val VAR2=val" "VAR=val" ...      ("-" = no variables)
out=$1; shift
mkdir -p $out
i=0
for v in "$@"; do
  i=$((i+1))
  if [ "$v" = "-" ]; then envs=""; else envs="$v"; fi
  env $envs python bench.py --config C5 --steps 2 --warmup 1 --no-cpu-baseline --no-host-path > $out/c$i.json 2> $out/c$i.err || { echo "variant '$v' FAILED"; tail -3 $out/c$i.err; continue; }
  python - "$out/c$i.json" "$v" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
st = d["roofline"]["stage_ms_per_step"]
print(f"{sys.argv[2]:40s} ms/step {d['ms_per_step']:8.2f}  " + "  ".join(f"{k} {v:.1f}" for k, v in st.items() if v), flush=True)
PY
done
