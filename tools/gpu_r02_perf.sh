#!/bin/bash
# round-2 GPU session for the block-stage kernels: per-mix memory ceilings, launch-size sweep with TLB / fabric counters,
# K2 / K3 / fused numbers, a sustained run, PMC traffic of the headline and of config 4
set -e
TAG=${1:-r02d}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/mem_ubench3 tools/ubench/mem_ubench3.hip
/tmp/mem_ubench3 > gpurun_out/${TAG}_mem_ubench3.txt 2>&1 || true
cat gpurun_out/${TAG}_mem_ubench3.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/mem_ubench2 tools/ubench/mem_ubench2.hip
/tmp/mem_ubench2 > gpurun_out/${TAG}_mem_ubench2.txt 2>&1 || true
python tools/bench_sustained.py > gpurun_out/${TAG}_sustained.json
cat gpurun_out/${TAG}_sustained.json
for f in 256 512 1024 2048 4096; do python bench.py --frames $f --steps 40 --no-cpu-baseline > gpurun_out/${TAG}_bench_f$f.json; done
grep -h -o '"frames_per_launch": [0-9]*\|"frac": [0-9.]*\|"kernel_ms": [0-9.]*' gpurun_out/${TAG}_bench_f*.json | paste - - -
python tools/bench_configs.py --config 2 > gpurun_out/${TAG}_k2.json
python tools/bench_configs.py --config 5 > gpurun_out/${TAG}_c5.json
python tools/bench_configs.py --config 7 > gpurun_out/${TAG}_c7.json
python tools/bench_configs.py --config 4 > gpurun_out/${TAG}_c4.json
cat gpurun_out/${TAG}_k2.json gpurun_out/${TAG}_c5.json gpurun_out/${TAG}_c7.json gpurun_out/${TAG}_c4.json | cut -c1-700
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > $ROOT/gpurun_out/${TAG}_counters_avail.txt 2>&1 || true
grep -i -o -E "\b(TCP_UTCL1[A-Z_0-9]*|TCP_TA[A-Z_0-9]*STALL[A-Z_0-9]*|TCC_EA[0-9]*_[A-Z_0-9]*|UTCL2[A-Z_0-9]*|TCC_TAG_STALL[A-Z_0-9]*)\b" $ROOT/gpurun_out/${TAG}_counters_avail.txt | sort -u | head -60 > $ROOT/gpurun_out/${TAG}_counters_tlb.txt || true
cat $ROOT/gpurun_out/${TAG}_counters_tlb.txt | tr '\n' ' '
