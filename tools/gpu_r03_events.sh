#!/bin/bash
# round-3 session: the batch pipelines with sleeping waits (poll + sleep, the default now) against spinning ones
# (HVC_EVENT_SPIN=1), same box, alternating: config 3 with the host reader (host-bound: a spinning orchestrator takes CPU
# time from the workers on a box with a CPU quota), config 3 with the GPU reader (upload-bound) and the file encoder
set -e
TAG=${1:-r03r}
mkdir -p gpurun_out
OUT=gpurun_out/${TAG}_events_ab.txt
: > $OUT
pick='"value": [0-9.]*\|"process_cpus_busy": [0-9.]*\|"verified": [a-z]*'
for round in 1 2 3; do
  for spin in 0 1; do
    echo "c3 host reader, 16 threads, spin=$spin: $(HVC_EVENT_SPIN=$spin python tools/bench_configs.py --config 3 --frames 1024 --threads 16 2>/dev/null | grep -o "$pick" | tr '\n' '\t')" | tee -a $OUT
  done
done
for round in 1 2; do
  for spin in 0 1; do
    echo "c3 GPU reader, 4096 files, spin=$spin: $(HVC_EVENT_SPIN=$spin python tools/bench_configs.py --config 3 --frames 4096 --threads 16 --gpu-entropy --chunk 256 2>/dev/null | grep -o "$pick" | tr '\n' '\t')" | tee -a $OUT
    echo "c5 files (encode batch), spin=$spin: $(HVC_EVENT_SPIN=$spin python tools/bench_configs.py --config 8 --threads 16 --gpu-entropy 2>/dev/null | grep -o '"value": [0-9.]*\|"verified": [a-z]*' | tr '\n' '\t')" | tee -a $OUT
  done
done
