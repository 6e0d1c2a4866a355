#!/bin/bash
# round-3 host-reader session: config 3 as BASELINE words it (host Huffman reader, 16 threads), the library as built
# against the same library with the PREVIOUS reader (build/variants/libhvc_oldreader.so: one symbol per refill, 37
# instructions a symbol), same box, alternating; then the tests that lean on the host reader.
# The other library is built once, here, from the commit before the rewrite (it is not kept in the tree):
#   git show c92bbc9:video-coding_amd/csrc/hvc_entropy.cpp > build/variants/old/hvc_entropy_old.cpp
#   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -pthread -Ivideo-coding_amd/csrc -c -o build/variants/old/hvc_entropy.o build/variants/old/hvc_entropy_old.cpp
#   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -pthread -shared -o build/variants/libhvc_oldreader.so build/obj/hvc_kernels.o build/obj/hvc_huff.o build/obj/hvc_hdec.o build/obj/hvc_capi*.o build/variants/old/hvc_entropy.o
# (tools/gpu_ab_libs.sh is the general form: any number of such builds against the library as built)
set -e
TAG=${1:-r03p}
mkdir -p gpurun_out
OUT=gpurun_out/${TAG}_c3_hostreader_ab.txt
: > $OUT
for round in 1 2 3; do
  for lib in new old; do
    for pairs in 1 0; do
      if [ $lib = old ]; then export HVC_JPEG_LIB=$PWD/build/variants/libhvc_oldreader.so; else unset HVC_JPEG_LIB; fi
      HVC_HOST_PAIRS=$pairs python tools/bench_configs.py --config 3 --frames 1024 --threads 16 > gpurun_out/${TAG}_c3_tmp.json 2>/dev/null
      echo "reader=$lib pairs=$pairs: $(grep -o '"value": [0-9.]*\|"entropy_Mpixel_s_per_thread": [0-9.]*\|"verified": [a-z]*' gpurun_out/${TAG}_c3_tmp.json | tr '\n' '\t')" | tee -a $OUT
    done
  done
done
unset HVC_JPEG_LIB
for t in 8 32; do
  python tools/bench_configs.py --config 3 --frames 1024 --threads $t > gpurun_out/${TAG}_c3_t$t.json 2>/dev/null
  echo "reader=new pairs=1 threads=$t: $(grep -o '"value": [0-9.]*\|"entropy_Mpixel_s_per_thread": [0-9.]*' gpurun_out/${TAG}_c3_t$t.json | tr '\n' '\t')" | tee -a $OUT
done
python tools/bench_configs.py --config 3 --frames 1024 --threads 16 > gpurun_out/${TAG}_c3.json 2>/dev/null
rm -f gpurun_out/${TAG}_c3_tmp.json
lscpu | grep -E "Model name|^CPU\(s\)|Thread|L1d" >> $OUT
python -m pytest tests/test_gpu_jpeg_api.py tests/test_gpu_fullsize_pipeline.py tests/test_gpu_concurrency.py tests/test_gpu_hdec.py -x -q -m gpu > gpurun_out/${TAG}_pytest_hostreader.log 2>&1 || true
tail -3 gpurun_out/${TAG}_pytest_hostreader.log
