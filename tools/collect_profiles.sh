#!/bin/bash
# One GPU session that produces everything profiles/<tag>/ holds.   usage: tools/collect_profiles.sh <tag> [quick]
# (run through gpurun; results land in gpurun_out/<tag>/ — copy what is to be judged into profiles/<tag>/;
#  profiles/pmc_traffic.json is updated in the snapshot and written to gpurun_out/<tag>/pmc_traffic.json)
set -u
TAG=${1:-prof}
QUICK=${2:-}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8   # what bench.py asks for; exported so that the profiler's early runtime start-up sees it too
cd $R
# counters first (one lane so that dispatches line up across passes): bench.py's roofline reads them from profiles/pmc_traffic.json
if [ -z "$QUICK" ]; then SETS=all; else SETS=core; fi
MCRT_LANES=1 PMC_SETS=$SETS $R/tools/pmc_run.sh gpurun_out/$TAG/pmc tools/gpu_case.py base 1 > $O/pmc_dispatches_1lane.txt 2>&1
python3 $R/tools/pmc_frame.py $O/pmc_dispatches_1lane.txt > $O/pmc_frame.txt 2>&1
python3 $R/tools/pmc_update.py 1080p_b4_spp4_S64 $O/pmc_dispatches_1lane.txt profiles/$TAG >> $O/pmc_update.log 2>&1
cp $O/pmc/summary.txt $O/pmc_summary_1lane.txt 2>/dev/null
# the metric frame once more in the launch shapes of a frame that shares the device (what `value` runs): the pipelined figures
MCRT_SHARED_GRIDS=1 MCRT_LANES=1 PMC_SETS=core $R/tools/pmc_run.sh gpurun_out/$TAG/pmc_shared tools/gpu_case.py base 1 > $O/pmc_dispatches_1lane_shared.txt 2>&1
python3 $R/tools/pmc_frame.py $O/pmc_dispatches_1lane_shared.txt > $O/pmc_frame_shared.txt 2>&1
python3 $R/tools/pmc_update.py 1080p_b4_spp4_S64 $O/pmc_dispatches_1lane_shared.txt profiles/$TAG pipelined >> $O/pmc_update.log 2>&1
rm -rf $O/pmc_shared/pass*
for PAIR in gui_defaults:gui 4k_b4_spp4_S64:4k_b4 4k_b8_spp16_S64:4k 8k_b8_spp64_S32:8k 256_b1_spp1_S64:c256; do
  W=${PAIR%%:*}; C=${PAIR##*:}
  MCRT_LANES=1 PMC_SETS=core $R/tools/pmc_run.sh gpurun_out/$TAG/pmc_$C tools/gpu_case.py $C 1 > $O/pmc_dispatches_$C.txt 2>&1
  python3 $R/tools/pmc_frame.py $O/pmc_dispatches_$C.txt > $O/pmc_frame_$C.txt 2>&1
  python3 $R/tools/pmc_update.py $W $O/pmc_dispatches_$C.txt profiles/$TAG >> $O/pmc_update.log 2>&1
  rm -rf $O/pmc_$C/pass*  # the raw per-pass CSVs are large; the dispatch tables keep what is used
  echo "pmc $W done"
done
cp $R/profiles/pmc_traffic.json $O/pmc_traffic.json
timeout -k 10 500 python3 bench.py --steps 50 --warmup 5 --check > $O/bench.json 2> $O/bench.err || echo "bench failed"
timeout -k 10 400 python3 bench.py --steps 50 --warmup 5 --frames-in-flight 1 --no-cpu-baseline --quick-host > $O/bench_one_frame_at_a_time.json 2>> $O/bench.err || echo "bench F=1 failed"
: > $O/other_workloads.jsonl
for W in gui_defaults 256_b1_spp1_S64 4k_b4_spp4_S64 4k_b8_spp16_S64 8k_b8_spp64_S32; do
  timeout -k 10 600 python3 bench.py --workload $W --steps 10 --warmup 2 --quick-host --check >> $O/other_workloads.jsonl 2>> $O/bench.err || echo "bench $W failed"
  echo "bench $W done"
done
cd /tmp
# the same bench command under the kernel tracer (four frames in flight: kernels of different frames overlap)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_bench -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --quick-host > $O/trace_bench.log 2>&1
python3 $R/tools/trace_summary.py $O/trace_bench 40 > $O/kernel_timeline_bench.txt 2>&1
# one lane: the plain dependency chain of a frame, kernel by kernel
MCRT_LANES=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_1lane -- python3 $R/tools/gpu_case.py base 10 > $O/trace_1lane.log 2>&1
python3 $R/tools/trace_summary.py $O/trace_1lane 9 > $O/kernel_timeline_1lane.txt 2>&1
# what the reference GUI renders by default (1080p, 64 spp, AO 16, depth of field): chain and bench line
MCRT_LANES=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_gui -- python3 $R/tools/gpu_case.py gui 5 > $O/trace_gui.log 2>&1
python3 $R/tools/trace_summary.py $O/trace_gui 12 > $O/kernel_timeline_gui_defaults.txt 2>&1
cp $O/trace_bench/*/*kernel_stats.csv $O/kernel_stats_bench.csv 2>/dev/null
cp $O/trace_1lane/*/*kernel_stats.csv $O/kernel_stats_1lane.csv 2>/dev/null
cp $O/trace_gui/*/*kernel_stats.csv $O/kernel_stats_gui_defaults.csv 2>/dev/null
rm -rf $O/pmc/pass*/ $O/trace_bench $O/trace_1lane $O/trace_gui
ls $O
cat $O/pmc_frame.txt
cat $O/bench.json
