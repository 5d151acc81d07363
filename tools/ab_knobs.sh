#!/bin/bash
# Same-box A/B of the library's development knobs: one bench.py line per setting.
#   usage (through gpurun): tools/ab_knobs.sh <out-file> "LABEL|ENV=1 ENV=2" ...
# columns: value (Mpixels/s, frames in flight), ms_per_step, latency_ms (a lone handle, library defaults), kernel.pipeline_ms
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$1; shift
mkdir -p "$(dirname "$OUT")"
WORKLOAD=${AB_WORKLOAD:-1080p_b4_spp4_S64}
STEPS=${AB_STEPS:-150}
for spec in "$@"; do
  label=${spec%%|*}; envs=${spec#*|}
  [ "$envs" = "$spec" ] && envs=""
  line=$(env $envs timeout -k 10 300 python3 $R/bench.py --no-cpu-baseline --steps $STEPS --workload $WORKLOAD ${AB_ARGS:-} 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'], d['latency_ms'], d['kernel']['pipeline_ms'], (d.get('render_call') or {}).get('ms'))")
  printf "%-34s %s\n" "$label" "$line" | tee -a "$OUT"
done
