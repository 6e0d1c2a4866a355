#!/bin/bash
# usage: tools/gpu_trace_cmd.sh <tag> <python script + args...>   -> gpurun_out/prof_<tag>/ (kernel trace only)
set -e
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python3 $ROOT/$@ > $OUT/trace.log 2>&1
