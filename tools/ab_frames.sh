#!/bin/bash
# same-box sweep of frames in flight x shared-mode grid size (development): value, ms_per_step
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=${1:-gpurun_out/final/frames_sweep.txt}
mkdir -p "$(dirname "$OUT")"
for F in 2 3 4 5 6 8; do
  for G in "" 768 1024 1536; do
    envs=""; [ -n "$G" ] && envs="MCRT_QUEUE_GRID=$G MCRT_PRIMARY_GRID=$G MCRT_RESOLVE_GRID=$G"
    line=$(env $envs timeout -k 10 200 python3 $R/bench.py --no-cpu-baseline --quick-host --steps 200 --frames-in-flight $F 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'])")
    printf "F=%s grid=%-6s %s\n" "$F" "${G:-auto}" "$line" | tee -a "$OUT"
  done
done
