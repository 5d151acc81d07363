#!/bin/bash
# Instruction counts and wave cycles of the last dispatch of each kernel.  usage: tools/pmc_valu.sh <tag> [case]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-v}; CASE=${2:-base}
export TMPDIR=/tmp
cd /tmp
MCRT_LANES=1 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $R/gpurun_out/$TAG/pmcv -- python3 $R/tools/gpu_case.py $CASE 1 > /dev/null 2>&1
MCRT_LANES=1 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d $R/gpurun_out/$TAG/pmcw -- python3 $R/tools/gpu_case.py $CASE 1 > /dev/null 2>&1
cd $R
python3 - <<PY
import csv, glob
last = {}
for f in glob.glob("gpurun_out/$TAG/pmc[vw]/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-28:]
        last.setdefault(k, {}).setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
for k, d in last.items():
    v = {}
    for disp in sorted(d): v.update(d[disp]) if len(d[disp]) else None
    v = d[max(d)]
    print(f"{k:30s}", " ".join(f"{n[3:]}={v[n]:.3e}" for n in sorted(v)))
PY
