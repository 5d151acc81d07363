#!/bin/bash
# same-box A/B of `plan_tiles`' waves per tile (development): bench lines per workload
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=${1:-gpurun_out/final/stream_waves.txt}
echo "== 1080p" | tee -a $OUT
AB_ARGS=--quick-host AB_STEPS=200 $R/tools/ab_knobs.sh $OUT "auto|" "waves1|MCRT_STREAM_WAVES=1" "waves4|MCRT_STREAM_WAVES=4" "auto2|"
for W in gui_defaults 4k_b8_spp16_S64 4k_b4_spp4_S64; do
  S=30; [ $W = 4k_b4_spp4_S64 ] && S=100
  echo "== $W" | tee -a $OUT
  AB_WORKLOAD=$W AB_ARGS=--quick-host AB_STEPS=$S $R/tools/ab_knobs.sh $OUT "auto|" "waves1|MCRT_STREAM_WAVES=1" "waves2|MCRT_STREAM_WAVES=2" "waves4|MCRT_STREAM_WAVES=4"
done
echo "== 8k" | tee -a $OUT
AB_WORKLOAD=8k_b8_spp64_S32 AB_ARGS=--quick-host AB_STEPS=6 $R/tools/ab_knobs.sh $OUT "auto|" "waves1|MCRT_STREAM_WAVES=1" "waves2|MCRT_STREAM_WAVES=2"
