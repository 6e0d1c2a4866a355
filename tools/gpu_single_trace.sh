#!/bin/bash
# the launches of one hvc_jpeg_decode call on one 1080p file (tools/trace_single_call.py under a kernel trace), with
# whatever environment the caller sets:   bash tools/gpu_single_trace.sh TAG [quality]
set -e
TAG=${1:-single}; Q=${2:-3}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
D=$ROOT/gpurun_out/prof_${TAG}_single; mkdir -p $D
(cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace -d $D -o trace -- python3 $ROOT/tools/trace_single_call.py --quality $Q > $D/log.txt 2>&1) || true
python3 tools/trace_single_call.py --timeline $D > gpurun_out/${TAG}_single_call_timeline.txt 2>&1 || true
tail -3 $D/log.txt
rm -rf $D
cat gpurun_out/${TAG}_single_call_timeline.txt
