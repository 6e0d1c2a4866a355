#!/bin/bash
# SQ counters of the reader's kernels on one 256-file chunk (tools/bench_reader_chunk.py), one counter group per run.
# usage: tools/gpu_pmc_reader.sh <tag> [VAR=value ...]
set -e
TAG=$1; shift
for a in "$@"; do export "$a"; done
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $ROOT/tools/bench_reader_chunk.py --files 256 --reps 2"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_BRANCH -d $OUT/pmc_insts -o pmc -- $CMD > $OUT/pmc_insts.log 2>&1 || true
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $OUT/pmc_active -o pmc -- $CMD > $OUT/pmc_active.log 2>&1 || true
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_INSTS_SMEM SQ_INST_LEVEL_LDS -d $OUT/pmc_lds -o pmc -- $CMD > $OUT/pmc_lds.log 2>&1 || true
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_CYCLES -d $OUT/pmc_misc -o pmc -- $CMD > $OUT/pmc_misc.log 2>&1 || true
cd $ROOT
python tools/rocpd_summary.py $OUT 0 > gpurun_out/${TAG}_rocprofv3.txt 2>&1 || true
python tools/reader_chunk_ms.py $OUT/trace >> gpurun_out/${TAG}_rocprofv3.txt 2>&1 || true
find $OUT -name '*.db' -delete
grep -A200 "pmc passes" gpurun_out/${TAG}_rocprofv3.txt | grep -E "k_hd_write2|k_hd_sync<" | head -80
