#!/bin/bash
# usage: tools/bench_variants.sh lib1.so lib2.so ...  — bench.py default line per variant build (value, ms_per_step, pipeline_ms)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for L in "$@"; do
  printf "%-28s " "$(basename $L)"
  MCRT_LIB=$R/$L timeout -k 10 200 python3 $R/bench.py --no-cpu-baseline --steps 150 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernel']['pipeline_ms'])"
done
