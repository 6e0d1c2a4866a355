#!/bin/bash
# same-box A/B of several builds of the host reader (build/variants/libhvc_<name>.so; "default" = the library as built),
# alternating: the reader alone on 1 and 16 threads, and config 3's pipeline
#   bash tools/gpu_ab_libs.sh TAG name1 name2 ...
set -e
TAG=$1; shift
mkdir -p gpurun_out
OUT=gpurun_out/${TAG}_reader_variants.txt
: > $OUT
for round in 1 2 3; do
  for name in default "$@"; do
    if [ $name = default ]; then unset HVC_JPEG_LIB; else export HVC_JPEG_LIB=$PWD/build/variants/libhvc_$name.so; fi
    a=$(python tools/bench_host_reader.py --threads 1,16 --seconds 1.5 2>/dev/null | grep '"two_files_in_turn": true' | grep -o '"Mpixel_s": [0-9.]*' | tr '\n' ' ')
    p=$(python tools/bench_configs.py --config 3 --frames 1024 --threads 16 2>/dev/null | grep -o '"value": [0-9.]*\|"verified": [a-z]*' | tr '\n' ' ')
    echo "reader=$name alone(1,16 threads): $a pipeline: $p" | tee -a $OUT
  done
done
