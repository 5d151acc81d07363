#!/bin/bash
# Collect rocprofv3 PMC counters in separate passes (no trace domains mixed in).
# usage: tools/pmc_run.sh <outdir> <python-script> [args...]     e.g. tools/pmc_run.sh gpurun_out/pmc bench.py --steps 3
#        PMC_SETS=core limits the run to the instruction-mix and HBM passes
set -u
OUT=$1; shift
SCRIPT=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp
mkdir -p "$R/$OUT"
cd /tmp
SETS=(
  "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES"
  "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES"
  "FETCH_SIZE"
  "WRITE_SIZE"
)
if [ "${PMC_SETS:-all}" = "all" ]; then
SETS+=(
  "SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INSTS_FLAT"
  "GRBM_GUI_ACTIVE GRBM_COUNT"
  "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES"
  "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_INT32"
)
fi
i=0
for SET in "${SETS[@]}"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --output-format csv -d "$R/$OUT/pass$i" -- python3 "$R/$SCRIPT" "$@" > "$R/$OUT/pass$i.log" 2>&1 || echo "pass $i failed (see pass$i.log)"
done
python3 "$R/tools/pmc_summarize.py" "$R/$OUT" > "$R/$OUT/summary.txt"
python3 "$R/tools/pmc_summarize.py" "$R/$OUT" --dispatches 64 | tee "$R/$OUT/dispatches.txt"
