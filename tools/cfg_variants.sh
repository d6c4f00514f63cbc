#!/bin/bash
# dev helper (GPU box): one bench line per environment variant for ANY --config (first argument), one summary line each
#   tools/cfg_variants.sh <config> <outdir> "VAR=val VAR2=val" "VAR=val" ...      ("-" = no variables)
cfg=$1; out=$2; shift; shift
mkdir -p $out
i=0
for v in "$@"; do
  i=$((i+1))
  if [ "$v" = "-" ]; then envs=""; else envs="$v"; fi
  env $envs python bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline --no-host-path > $out/g$i.json 2> $out/g$i.err || { echo "variant '$v' FAILED"; tail -3 $out/g$i.err; continue; }
  python - "$out/g$i.json" "$v" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
st = d["roofline"]["stage_ms_per_step"]
c = d["config"]
print(f"{sys.argv[2]:32s} ms/step {d['ms_per_step']:8.2f}  " + "  ".join(f"{k} {v:.2f}" for k, v in st.items() if v) +
      f"  | planes {c['w_planes']} W {c['kernel_support']} grid {c['grid'][0]} {c['w_scheme'][:4]}", flush=True)
PY
done
