#!/bin/bash
# per-file Huffman tables (PF mode): parity, then the reader alone on one 256-file chunk and the pipeline at 4096 files
set -e
TAG=${1:-r02y}
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_hdec.py tests/test_gpu_jpeg_api.py -m gpu -q -x 2>&1 | tail -2
D=gpurun_out/prof_${TAG}_chunk_own; mkdir -p $D
(cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats -d $OLDPWD/$D/trace -o trace -- python3 $OLDPWD/tools/bench_reader_chunk.py --files 256 --reps 4 --own-tables > $OLDPWD/$D/trace.log 2>&1)
{ grep records_equal $D/trace.log; python tools/reader_chunk_ms.py $D/trace; } | tee gpurun_out/${TAG}_reader_chunk_own_tables.txt
find $D -name '*.db' -delete
python tools/bench_configs.py --config 3 --frames 4096 --threads 16 --gpu-entropy --chunk 256 --own-tables 2>/dev/null | tee gpurun_out/${TAG}_c3g_4096_own.json | grep -o '"verified": [a-z]*\|"value": [0-9.]*' | paste - -
python tools/stress_hdec.py --cases 150 --mutations 200 --seed 61 2>&1 | tail -1
