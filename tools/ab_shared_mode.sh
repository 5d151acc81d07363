for W in 1080p_b4_spp4_S64 4k_b4_spp4_S64 4k_b8_spp16_S64 gui_defaults 256_b1_spp1_S64; do
  S=150; [ $W = 4k_b8_spp16_S64 ] && S=40; [ $W = gui_defaults ] && S=30
  echo "== $W" | tee -a gpurun_out/final/shared_mode.txt
  AB_WORKLOAD=$W AB_ARGS=--quick-host AB_STEPS=$S tools/ab_knobs.sh gpurun_out/final/shared_mode.txt "auto|" "solo|MCRT_SHARED_GRIDS=0" "shared|MCRT_SHARED_GRIDS=1" "auto2|"
done
echo "== 8k" | tee -a gpurun_out/final/shared_mode.txt
AB_WORKLOAD=8k_b8_spp64_S32 AB_ARGS=--quick-host AB_STEPS=6 tools/ab_knobs.sh gpurun_out/final/shared_mode.txt "auto|" "solo|MCRT_SHARED_GRIDS=0" "shared|MCRT_SHARED_GRIDS=1"
