#!/bin/bash
# A/B of library variants on the file-level pipeline (config 3, GPU Huffman reader): throughput at 4096 files and the GPU
# work per 256-file chunk from a kernel trace.   usage: tools/gpu_ab_reader.sh <tag> <name>=<library>[@VAR=value] ...
set -e
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
OUTF=$ROOT/gpurun_out/${TAG}_ab.txt
: > $OUTF
for spec in "$@"; do
    name=${spec%%=*}; lib=${spec#*=}
    unset HVC_WR_MODE
    case $lib in *@*) export "${lib#*@}"; lib=${lib%%@*};; esac
    export HVC_JPEG_LIB=$ROOT/$lib
    echo "== $name ($lib)" | tee -a $OUTF
    timeout -k 10 300 python -m pytest tests/test_gpu_hdec.py -m gpu -q -x 2>&1 | tail -1 | tee -a $OUTF
    for rep in 1 2; do
        python tools/bench_configs.py --config 3 --frames 4096 --threads 16 --gpu-entropy --chunk 256 2>/dev/null | grep -o '"verified": [a-z]*\|"value": [0-9.]*\|"kernel_ms_sum": [0-9.]*' | paste - - - | tee -a $OUTF
    done
    D=$ROOT/gpurun_out/prof_${TAG}_$name
    mkdir -p $D
    # the reader alone on one 256-file chunk, one stream: per-kernel durations without the other reader stream in them
    (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats -d $D/trace -o trace -- python3 $ROOT/tools/bench_reader_chunk.py --files 256 --reps 4 > $D/trace.log 2>&1)
    grep records_equal $D/trace.log | tee -a $OUTF
    python tools/reader_chunk_ms.py $D/trace | tee -a $OUTF
    find $D -name '*.db' -delete
done
