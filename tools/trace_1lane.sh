#!/bin/bash
# Kernel timeline of one frame on one stream.   usage (through gpurun): tools/trace_1lane.sh <tag> [case] [iters]
set -u
TAG=${1:-t}; CASE=${2:-base}; ITERS=${3:-10}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
MCRT_LANES=${MCRT_LANES:-1} rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$CASE -- python3 $R/tools/gpu_case.py $CASE $ITERS > $O/trace_$CASE.log 2>&1
python3 $R/tools/trace_summary.py $O/trace_$CASE ${4:-12} | tee $O/kernel_timeline_$CASE.txt
