#!/bin/bash
# round-2 GPU session for the config-3 pipeline: tests, then the reader's throughput in its variants, then a kernel trace
set -e
TAG=${1:-r02c}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/${TAG}_pytest.log 2>&1 || { tail -30 gpurun_out/${TAG}_pytest.log; exit 1; }
tail -3 gpurun_out/${TAG}_pytest.log
C3="python tools/bench_configs.py --config 3 --threads 16 --gpu-entropy --chunk 256 --steps 3"
$C3 --frames 1024 > gpurun_out/${TAG}_c3g_1024.json
$C3 --frames 4096 > gpurun_out/${TAG}_c3g_4096.json
$C3 --frames 4096 --own-tables > gpurun_out/${TAG}_c3g_4096_own.json
for n in 6 8 10; do HVC_HD_SYNC_ROUNDS=$n $C3 --frames 4096 > gpurun_out/${TAG}_c3g_4096_rounds$n.json; done
python tools/bench_configs.py --config 3 --threads 16 --frames 1024 > gpurun_out/${TAG}_c3_host.json
grep -h -o '"config": "[^"]*"\|"value": [0-9.]*\|"verified": [a-z]*' gpurun_out/${TAG}_c3*.json | paste - - - 
bash tools/gpu_profile_cmd.sh ${TAG}_c3g tools/bench_configs.py --config 3 --frames 512 --threads 16 --gpu-entropy --chunk 256 --steps 2
python tools/rocpd_summary.py gpurun_out/prof_${TAG}_c3g > gpurun_out/${TAG}_c3g_rocprofv3.txt 2>&1 || true
find gpurun_out/prof_${TAG}_c3g -name '*.db' -delete
head -30 gpurun_out/${TAG}_c3g_rocprofv3.txt
