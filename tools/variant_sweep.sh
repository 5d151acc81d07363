#!/bin/bash
# usage: tools/variant_sweep.sh <case> <iters> lib1.so lib2.so ...   — kernel-only ms per variant build
# (each lib may be prefixed with ENV=VAL, ... e.g. MCRT_QUEUE_GRID=4096:variants/x.so)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
C=$1; shift; N=$1; shift
for A in "$@"; do
  L=${A##*:}; E=""; [ "$A" != "$L" ] && E=${A%:*}
  printf "%-44s " "$A"
  env $E MCRT_LIB=$R/$L timeout -k 10 120 python3 $R/tools/gpu_case.py $C $N 2>/dev/null | tail -1
done
