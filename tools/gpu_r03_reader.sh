#!/bin/bash
# round-3 reader session: the new tests (overflow search, photograph-like content), then the reader alone on one 256-file
# chunk under a kernel trace (per-kernel times: did the overflow branch cost the hot loops anything?) and the pipeline
set -e
TAG=${1:-r03h}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
python -m pytest tests/test_gpu_hdec.py tests/test_gpu_photo_content.py -x -q -m gpu -s > gpurun_out/${TAG}_pytest.log 2>&1 || true
tail -5 gpurun_out/${TAG}_pytest.log
for v in "" "--own-tables"; do
  n=reader_chunk$(echo "$v" | sed 's/--own-tables/_own_tables/')
  D=$ROOT/gpurun_out/prof_${TAG}_$n; mkdir -p $D
  (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats -d $D/trace -o trace -- python3 $ROOT/tools/bench_reader_chunk.py --files 256 --reps 4 $v > $D/trace.log 2>&1) || true
  { grep records_equal $D/trace.log; python tools/reader_chunk_ms.py $D/trace; } > gpurun_out/${TAG}_$n.txt 2>&1 || true
  rm -rf $D
  cat gpurun_out/${TAG}_$n.txt
done
python tools/bench_configs.py --config 3 --frames 4096 --threads 16 --gpu-entropy --chunk 256 > gpurun_out/${TAG}_c3_gpu_entropy_4096.json 2>/dev/null
python tools/bench_configs.py --config 3 --frames 4096 --threads 16 --gpu-entropy --chunk 256 --own-tables > gpurun_out/${TAG}_c3_gpu_entropy_4096_own_tables.json 2>/dev/null
grep -o '"value": [0-9.]*\|"verified": [a-z]*' gpurun_out/${TAG}_c3_gpu_entropy_4096*.json
