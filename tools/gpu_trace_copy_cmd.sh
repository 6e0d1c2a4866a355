#!/bin/bash
# usage: tools/gpu_trace_copy_cmd.sh <tag> <python script + args...>   -> gpurun_out/prof_<tag>/ (kernel + memory-copy trace)
set -e
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace -d $OUT/trace -o trace -- python3 $ROOT/$@ > $OUT/trace.log 2>&1
