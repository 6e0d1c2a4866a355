#!/bin/bash
# soak of the file-level pipeline with the GPU Huffman reader: the same 4096-file batches over and over (model's tables,
# per-file tables, host output, 4:4:4 output where the tool has it), every output checksummed on the device against the
# golden values -- a race in the reader's lists or in the pipeline's slot hand-over shows up as "verified": false
set -e
N=${1:-8}
mkdir -p gpurun_out
: > gpurun_out/soak.txt
for i in $(seq 1 $N); do
  for v in "" "--own-tables" "--host-out"; do
    python tools/bench_configs.py --config 3 --frames 4096 --threads 16 --gpu-entropy --chunk 256 --steps 3 $v 2>/dev/null | grep -o '"verified": [a-z]*\|"value": [0-9.]*' | paste - - | sed "s/^/run $i $v: /" | tee -a gpurun_out/soak.txt
  done
done
grep -c '"verified": true' gpurun_out/soak.txt
! grep -q '"verified": false' gpurun_out/soak.txt
