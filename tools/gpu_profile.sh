#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + PMC passes of bench.py.
# usage: tools/gpu_profile.sh <tag> [bench args...]   -> gpurun_out/prof_<tag>/
set -e
TAG=${1:-r01}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 40 --warmup 10 --no-cpu-baseline --no-others --no-other-layout --sustain-seconds 2 $@"   # (the profile is of the headline step alone)
# the kernel trace is of the command exactly as the driver runs it (`python bench.py`, or with the caller's arguments): the
# dominant kernel's average over its TIMED dispatches (tools/rocpd_summary.py) is what bench.py's own event time must agree with
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python3 $ROOT/bench.py "$@" > $OUT/trace_bench.log 2>&1
# PMC passes (own runs; no trace domains)
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc -- python3 $ROOT/bench.py $ARGS > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o pmc -- python3 $ROOT/bench.py $ARGS > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $OUT/pmc_sq -o pmc -- python3 $ROOT/bench.py $ARGS > $OUT/pmc_sq.log 2>&1 || true
rocprofv3 --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc_misc -o pmc -- python3 $ROOT/bench.py $ARGS > $OUT/pmc_misc.log 2>&1 || true
find $OUT -name "*.csv" | head -50
