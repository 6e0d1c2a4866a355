#!/bin/bash
# usage: tools/gpu_profile_cmd.sh <tag> <python script + args...>   -> gpurun_out/prof_<tag>/  (kernel trace + SQ/TCC PMC passes)
set -e
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python3 $ROOT/$@ > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc -- python3 $ROOT/$@ > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o pmc -- python3 $ROOT/$@ > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $OUT/pmc_sq -o pmc -- python3 $ROOT/$@ > $OUT/pmc_sq.log 2>&1 || true
rocprofv3 --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc_misc -o pmc -- python3 $ROOT/$@ > $OUT/pmc_misc.log 2>&1 || true
