#!/bin/bash
# VALU / SALU / LDS instruction counts and wave cycles of the last dispatch of each kernel.  usage: tools/pmc_valu.sh <tag> [case]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-v}; CASE=${2:-base}
export TMPDIR=/tmp
cd /tmp
MCRT_LANES=1 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $R/gpurun_out/$TAG/pmcv -- python3 $R/tools/gpu_case.py $CASE 1 > /dev/null 2>&1
cd $R
python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/$TAG/pmcv/*/*counter_collection.csv")[0]
last = {}
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0][-28:]
    last.setdefault(k, {}).setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
for k, d in last.items():
    v = d[max(d)]
    print(f"{k:30s}", " ".join(f"{n[3:]}={v[n]:.3e}" for n in sorted(v)))
PY
