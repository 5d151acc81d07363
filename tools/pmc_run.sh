#!/bin/bash
# Collect rocprofv3 PMC counters in separate passes (no trace domains mixed in).
# usage: tools/pmc_run.sh <outdir> <python-script> [args...]     e.g. tools/pmc_run.sh gpurun_out/pmc bench.py --steps 3
set -u
OUT=$1; shift
SCRIPT=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp
mkdir -p "$R/$OUT"
cd /tmp
i=0
for SET in \
  "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES" \
  "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" \
  "SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INSTS_FLAT" \
  "FETCH_SIZE" \
  "WRITE_SIZE" \
  "GRBM_GUI_ACTIVE GRBM_COUNT" \
  "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES" \
  "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_INT32" ; do
  i=$((i+1))
  rocprofv3 --pmc $SET --output-format csv -d "$R/$OUT/pass$i" -- python3 "$R/$SCRIPT" "$@" > "$R/$OUT/pass$i.log" 2>&1 || echo "pass $i failed (see pass$i.log)"
done
python3 "$R/tools/pmc_summarize.py" "$R/$OUT" | tee "$R/$OUT/summary.txt"
