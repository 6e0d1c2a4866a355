#!/bin/bash
# One GPU session that produces every number DESIGN.md quotes (run via gpurun from the repo root):
#   gpurun_out/m_*.json  bench lines, gpurun_out/prof_<tag>/ rocprofv3 databases
set -e
TAG=${1:-r01}
mkdir -p gpurun_out
python bench.py > gpurun_out/m_bench.json 2> gpurun_out/m_bench.err
echo "bench done"
python tools/bench_configs.py --config 3 --frames 1024 --threads 16 > gpurun_out/m_c3.json 2> gpurun_out/m_c3.err
echo "c3 done"
python tools/bench_configs.py --config 4 > gpurun_out/m_c4.json 2> gpurun_out/m_c4.err
python tools/bench_configs.py --config 5 > gpurun_out/m_c5.json 2> gpurun_out/m_c5.err
python tools/bench_configs.py --config 6 > gpurun_out/m_c6.json 2> gpurun_out/m_c6.err
python tools/bench_configs.py --config 7 > gpurun_out/m_c7.json 2> gpurun_out/m_c7.err
python tools/bench_configs.py --config 2 > gpurun_out/m_k2.json 2> gpurun_out/m_k2.err
python tools/bench_configs.py --config 8 > gpurun_out/m_c5_files.json 2> gpurun_out/m_c5_files.err
HVC_DECODE_KERNEL=q16 python bench.py --no-cpu-baseline > gpurun_out/m_bench_q16.json 2> gpurun_out/m_bench_q16.err
# the file-level pipelines with the entropy stages on the GPU, the GPU Huffman coder alone, one file at a time
python tools/bench_configs.py --config 3 --frames 1024 --threads 16 --gpu-entropy --chunk 64 > gpurun_out/m_c3_gpu_entropy.json 2> gpurun_out/m_c3g.err
python tools/bench_configs.py --config 3 --frames 4096 --threads 16 --gpu-entropy --chunk 256 > gpurun_out/m_c3_gpu_entropy_4096.json 2>> gpurun_out/m_c3g.err
python tools/bench_configs.py --config 8 --gpu-entropy > gpurun_out/m_c5_files_gpu_entropy.json 2> gpurun_out/m_c5g.err
python tools/bench_configs.py --config 9 > gpurun_out/m_huffman_gpu.json 2> gpurun_out/m_huffman_gpu.err
python tools/bench_single.py > gpurun_out/m_single_file.jsonl 2> gpurun_out/m_single_file.err
echo "configs done"
bash tools/gpu_profile.sh ${TAG}_decode
echo "decode profile done"
bash tools/gpu_profile_cmd.sh ${TAG}_444 tools/bench_configs.py --config 7 --steps 10
echo "444 profile done"
bash tools/gpu_profile_cmd.sh ${TAG}_encode tools/bench_configs.py --config 5 --steps 10
echo "encode profile done"
bash tools/gpu_profile_cmd.sh ${TAG}_c3g tools/bench_configs.py --config 3 --frames 512 --threads 16 --gpu-entropy --chunk 256 --steps 2
echo "reader profile done"
# summaries here, databases deleted: gpurun only brings back 64 MiB
for d in gpurun_out/prof_${TAG}_*; do
    python tools/rocpd_summary.py $d > gpurun_out/$(basename $d | sed 's/^prof_//')_rocprofv3.txt 2>&1 || true
    find $d -name '*.db' -delete
done
echo "summaries done"
