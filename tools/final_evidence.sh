O=gpurun_out/r03_v5b; mkdir -p $O
export GPU_MAX_HW_QUEUES=8
timeout -k 10 500 python3 bench.py --steps 50 --warmup 5 --check > $O/bench.json 2> $O/bench.err || echo "bench failed"
timeout -k 10 400 python3 bench.py --steps 50 --warmup 5 --frames-in-flight 1 --no-cpu-baseline --quick-host > $O/bench_one_frame_at_a_time.json 2>> $O/bench.err || echo "bench F=1 failed"
: > $O/other_workloads.jsonl
for W in gui_defaults 256_b1_spp1_S64 4k_b4_spp4_S64 4k_b8_spp16_S64 8k_b8_spp64_S32; do
  timeout -k 10 600 python3 bench.py --workload $W --steps 10 --warmup 2 --quick-host --check >> $O/other_workloads.jsonl 2>> $O/bench.err || echo "bench $W failed"
  echo "bench $W done"
done
timeout -k 10 300 python3 tools/gpu_fuzz.py 70000000 8000 > $O/fuzz_general_8000.txt 2>&1; tail -1 $O/fuzz_general_8000.txt
timeout -k 10 250 python3 tools/gpu_fuzz.py 71000000 6000 bundle > $O/fuzz_bundle_6000.txt 2>&1; tail -1 $O/fuzz_bundle_6000.txt
timeout -k 10 250 python3 tools/gpu_fuzz.py 72000000 6000 wide > $O/fuzz_wide_6000.txt 2>&1; tail -1 $O/fuzz_wide_6000.txt
MCRT_SLAB_MIN_SPP=2 timeout -k 10 200 python3 tools/gpu_fuzz.py 73000000 3000 > $O/fuzz_slab_everywhere_3000.txt 2>&1; tail -1 $O/fuzz_slab_everywhere_3000.txt
MCRT_SHARED_GRIDS=1 timeout -k 10 200 python3 tools/gpu_fuzz.py 74000000 3000 > $O/fuzz_shared_device_shapes_3000.txt 2>&1; tail -1 $O/fuzz_shared_device_shapes_3000.txt
MCRT_STREAM_WAVES=2 timeout -k 10 200 python3 tools/gpu_fuzz.py 75000000 2000 > $O/fuzz_two_stream_waves_2000.txt 2>&1; tail -1 $O/fuzz_two_stream_waves_2000.txt
timeout -k 10 500 tools/decide_check.sh run 76000000 1500 > $O/decide_check.txt 2>&1; cat $O/decide_check.txt
