#!/bin/bash
# SQ_INSTS_VALU / SALU of the level-0 shadow dispatch for a list of variant builds
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp; cd /tmp
for L in "$@"; do
  D=$R/gpurun_out/abl_$(basename $L .so)
  MCRT_LANES=1 MCRT_GRAPH=0 MCRT_LIB=$R/$L rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --output-format csv -d $D -- python3 $R/tools/gpu_case.py base 1 > $D.log 2>&1
  python3 - "$D" "$L" <<'PY'
import csv, glob, sys
from collections import defaultdict
rows = defaultdict(dict); names = {}; order = {}
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        d = int(r["Dispatch_Id"]); order.setdefault(d, len(order)); i = order[d]
        rows[i][r["Counter_Name"]] = rows[i].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"]); names[i] = r["Kernel_Name"].split("(")[0]
first = [i for i in sorted(rows) if i < 16]
print(sys.argv[2])
for i in first:
    if any(k in names[i] for k in ("shadow", "shade", "primary", "light")):
        print(f"   {i:2d} {names[i][-34:]:34s} VALU={rows[i].get('SQ_INSTS_VALU',0):.4g} SALU={rows[i].get('SQ_INSTS_SALU',0):.4g}")
PY
done
