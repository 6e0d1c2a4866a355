#!/bin/bash
# usage: tools/gpu_pmc_cmd.sh <tag> "<counters>" <python script + args...>   -> gpurun_out/pmc_<tag>/ (one PMC pass)
set -e
TAG=$1; shift
CTR=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 150 rocprofv3 --pmc $CTR -d $OUT/pmc_sq -o pmc -- python3 $ROOT/$@ > $OUT/pmc.log 2>&1
