#!/bin/bash
# same-box A/B of the host reader: the library as built against build/variants/$1 (a library with another reader),
# alternating -- the reader alone on 1 and 16 threads (tools/bench_host_reader.py) and config 3's pipeline
set -e
OTHER=$PWD/build/variants/$1
TAG=${2:-ab}
mkdir -p gpurun_out
OUT=gpurun_out/${TAG}_reader_ab.txt
: > $OUT
for round in 1 2 3; do
  for lib in new other; do
    if [ $lib = other ]; then export HVC_JPEG_LIB=$OTHER; else unset HVC_JPEG_LIB; fi
    python tools/bench_host_reader.py --threads 1,16 --seconds 1.5 2>/dev/null | grep '"two_files_in_turn": true' | sed "s/^/reader=$lib alone: /" | tee -a $OUT
    echo "reader=$lib pipeline: $(python tools/bench_configs.py --config 3 --frames 1024 --threads 16 2>/dev/null | grep -o '"value": [0-9.]*\|"entropy_Mpixel_s_per_thread": [0-9.]*\|"verified": [a-z]*' | tr '\n' '\t')" | tee -a $OUT
  done
done
