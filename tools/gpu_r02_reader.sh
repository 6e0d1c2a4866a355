#!/bin/bash
# round-2 reader session: parity of the Huffman reader kernels after a change, then the file-level pipeline
# (config 3, 4096 files, chunks of 256) with the default and with per-file tables, and a kernel trace of both.
set -e
TAG=${1:-r02k}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_hdec.py tests/test_gpu_jpeg_api.py tests/test_gpu_fullsize_pipeline.py -m gpu -q -x > gpurun_out/${TAG}_pytest.log 2>&1 || { tail -40 gpurun_out/${TAG}_pytest.log; exit 1; }
tail -2 gpurun_out/${TAG}_pytest.log
for rep in 1 2; do
python tools/bench_configs.py --config 3 --frames 4096 --threads 16 --gpu-entropy --chunk 256 2>/dev/null | tee -a gpurun_out/${TAG}_c3g_4096.json
done
python tools/bench_configs.py --config 3 --frames 4096 --threads 16 --gpu-entropy --chunk 256 --own-tables 2>/dev/null | tee gpurun_out/${TAG}_c3g_4096_own.json
python tools/bench_single.py 2>/dev/null | tee gpurun_out/${TAG}_single_file.jsonl
# the reader alone on one 256-file chunk, one stream (per-kernel durations that do not depend on the other reader stream)
D=gpurun_out/prof_${TAG}_chunk; mkdir -p $D
(cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats -d $OLDPWD/$D/trace -o trace -- python3 $OLDPWD/tools/bench_reader_chunk.py --files 256 --reps 4 > $OLDPWD/$D/trace.log 2>&1)
{ grep records_equal $D/trace.log; python tools/reader_chunk_ms.py $D/trace; } | tee gpurun_out/${TAG}_reader_chunk.txt
D=gpurun_out/prof_${TAG}_chunk_own; mkdir -p $D
(cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats -d $OLDPWD/$D/trace -o trace -- python3 $OLDPWD/tools/bench_reader_chunk.py --files 256 --reps 4 --own-tables > $OLDPWD/$D/trace.log 2>&1)
{ grep records_equal $D/trace.log; python tools/reader_chunk_ms.py $D/trace; } | tee gpurun_out/${TAG}_reader_chunk_own_tables.txt
find gpurun_out/prof_${TAG}_chunk gpurun_out/prof_${TAG}_chunk_own -name '*.db' -delete
bash tools/gpu_profile_cmd.sh ${TAG}_c3g tools/bench_configs.py --config 3 --frames 512 --threads 16 --gpu-entropy --chunk 256 --steps 2 > /dev/null 2>&1
python tools/rocpd_summary.py gpurun_out/prof_${TAG}_c3g 10 > gpurun_out/${TAG}_c3g_rocprofv3.txt 2>&1 || true
find gpurun_out/prof_${TAG}_c3g -name '*.db' -delete
grep -E "k_hd_|k_decode_packed" gpurun_out/${TAG}_c3g_rocprofv3.txt | head -8
