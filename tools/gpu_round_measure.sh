#!/bin/bash
# One GPU session that produces every number DESIGN.md section 5 quotes (run via gpurun from the repo root), limited to SURVEY.md
# section 8's rows (VERDICT r4's stop list: no single-file, small-batch, restart-interval, convert or reader-chunk figures):
#   PART=a  the bench lines:      gpurun_out/<tag>_lines.jsonl (every JSON line, tagged with its command's name: "what"),
#                                 <tag>_stderr.txt, <tag>_shape_ceilings.txt, <tag>_mem_ubench3.txt
#   PART=b  the rocprofv3 passes: gpurun_out/<tag>_{decode,decode_c4,decode_c3,encode,444,wide}_rocprofv3.txt (kernel trace + PMC averages) and
#                                 gpurun_out/traffic_<tag>.jsonl (one profiles/traffic.json entry per line)
# The driver's own two commands (the GPU suite and smoke()) are NOT in here: they are issued as direct gpurun calls.
set -e
TAG=${1:-r05}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
PART=${PART:-all}
if [ "$PART" != "b" ]; then
python bench.py > gpurun_out/m_bench.json 2> gpurun_out/m_bench.err
python bench.py --config 4 --no-cpu-baseline > gpurun_out/m_bench_c4.json 2> gpurun_out/m_bench_c4.err
python bench.py --config 3 > gpurun_out/m_bench_c3.json 2> gpurun_out/m_bench_c3.err
python bench.py --config 5 > gpurun_out/m_bench_c5.json 2> gpurun_out/m_bench_c5.err
python bench.py --frames 4096 --steps 20 --no-cpu-baseline --no-others > gpurun_out/m_bench_f4096.json 2> /dev/null
HVC_DECODE_KERNEL=q16 python bench.py --no-cpu-baseline --no-others --sustain-seconds 0 > gpurun_out/m_bench_q16.json 2> gpurun_out/m_bench_q16.err
echo "bench done"
# N-rank rehearsals on this one GPU (all ranks on cuda:0, gloo): never measurements, the lines say so
HVC_BENCH_REHEARSAL=1 python bench.py --gpus 2 --steps 20 --no-cpu-baseline 2> gpurun_out/m_bench_rehearsal2.err | grep '^{' > gpurun_out/m_bench_rehearsal2.json
HVC_BENCH_REHEARSAL=1 python bench.py --gpus 4 --config 4 --steps 1 --warmup 0 --frames 16 --shard 64 --no-cpu-baseline --sustain-seconds 0 2> gpurun_out/m_bench_rehearsal4_c4.err | grep '^{' > gpurun_out/m_bench_rehearsal4_c4.json
HVC_BENCH_REHEARSAL=1 python bench.py --gpus 2 --config 3 2> gpurun_out/m_bench_rehearsal2_c3.err | grep '^{' > gpurun_out/m_bench_rehearsal2_c3.json
HVC_BENCH_REHEARSAL=1 python bench.py --gpus 2 --config 5 --steps 10 2> gpurun_out/m_bench_rehearsal2_c5.err | grep '^{' > gpurun_out/m_bench_rehearsal2_c5.json
echo "rehearsals done"
python tools/bench_configs.py --config 7 > gpurun_out/m_c7.json 2> gpurun_out/m_c7.err                       # fused 4:4:4 against its three-launch composition
python tools/bench_configs.py --config 2 > gpurun_out/m_k2.json 2> gpurun_out/m_k2.err                       # K2 alone
python tools/bench_configs.py --config 10 > gpurun_out/m_sub420.json 2> gpurun_out/m_sub420.err              # subsample_hv2
python tools/bench_configs.py --config 6 > gpurun_out/m_c6.json 2> gpurun_out/m_c6.err                       # host-buffer boundary (PCIe-inclusive)
python tools/bench_configs.py --config 8 > gpurun_out/m_c5_files.json 2> gpurun_out/m_c5_files.err           # config 5 to files, host coder
python tools/bench_configs.py --config 8 --gpu-entropy > gpurun_out/m_c5_files_gpu_entropy.json 2> gpurun_out/m_c5g.err
python tools/bench_configs.py --config 9 > gpurun_out/m_huffman_gpu.json 2> gpurun_out/m_huffman_gpu.err      # GPU Huffman coder alone
python tools/bench_configs.py --config 12 --threads 16 > gpurun_out/m_seam.json 2> gpurun_out/m_seam.err      # the asynchronous seam: pinned slots -> HBM, 4096 records
python tools/bench_configs.py --config 12 --threads 16 --host-out --frames 2048 > gpurun_out/m_seam_host.json 2> gpurun_out/m_seam_host.err   # ... pixels back into pinned slots
python tools/bench_configs.py --config 13 > gpurun_out/m_wide.json 2> gpurun_out/m_wide.err                  # every block through the int64 kernel
python tools/bench_configs.py --config 13 --wide-mode dqt16 > gpurun_out/m_wide_dqt16.json 2> gpurun_out/m_wide_dqt16.err
python tools/bench_configs.py --config 14 > gpurun_out/m_fixup.json 2> gpurun_out/m_fixup.err                 # every block through the fix-up list (adversarial)
echo "configs done"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/mem_ubench3 tools/ubench/mem_ubench3.hip 2> /dev/null
/tmp/mem_ubench3 > gpurun_out/m_mem_ubench3.txt 2>&1 || true
bash tools/gpu_shape_ceilings.sh > gpurun_out/m_shape_ceilings.txt 2> /dev/null || true
echo "ceilings done"
python tools/consolidate_session.py gpurun_out m ${TAG}
fi
if [ "$PART" = "a" ]; then exit 0; fi
bash tools/gpu_profile.sh ${TAG}_decode
echo "decode profile done"
bash tools/gpu_profile.sh ${TAG}_decode_c4 --config 4 --steps 2 --warmup 1
echo "config-4 profile done"
bash tools/gpu_profile_cmd.sh ${TAG}_444 tools/bench_configs.py --config 7 --steps 40 --warmup 20 --fused-only
echo "444 profile done"
bash tools/gpu_profile_cmd.sh ${TAG}_encode tools/bench_configs.py --config 5 --steps 40 --warmup 20
echo "encode profile done"
# config 3's host-reader pipeline: its k_decode_packed launches are chunks of 32 frames (also the warm-up call's)
bash tools/gpu_profile_cmd.sh ${TAG}_decode_c3 tools/bench_configs.py --config 3 --frames 1024 --steps 1 --threads 16 --chunk 32
echo "config-3 profile done"
bash tools/gpu_profile_cmd.sh ${TAG}_wide tools/bench_configs.py --config 13 --steps 20 --warmup 5
echo "wide profile done"
: > gpurun_out/traffic_${TAG}.jsonl
python tools/rocpd_summary.py gpurun_out/prof_${TAG}_decode --traffic k_decode_packed 1024 2 ${TAG} | tr -d '\n' >> gpurun_out/traffic_${TAG}.jsonl; echo >> gpurun_out/traffic_${TAG}.jsonl
python tools/rocpd_summary.py gpurun_out/prof_${TAG}_decode_c4 --traffic k_decode_packed 128 4 ${TAG} | tr -d '\n' >> gpurun_out/traffic_${TAG}.jsonl; echo >> gpurun_out/traffic_${TAG}.jsonl
python tools/rocpd_summary.py gpurun_out/prof_${TAG}_encode --traffic k_encode 256 5 ${TAG} | tr -d '\n' >> gpurun_out/traffic_${TAG}.jsonl; echo >> gpurun_out/traffic_${TAG}.jsonl
python tools/rocpd_summary.py gpurun_out/prof_${TAG}_444 --traffic 'k_decode_444<' 512 7 ${TAG} | tr -d '\n' >> gpurun_out/traffic_${TAG}.jsonl; echo >> gpurun_out/traffic_${TAG}.jsonl
python tools/rocpd_summary.py gpurun_out/prof_${TAG}_decode_c3 --traffic k_decode_packed 32 3 ${TAG} "tools/bench_configs.py --config 3 --chunk 32: the host-reader pipeline of bench.py --config 3 (hvc_jpeg_decode_batch), 32-frame launches" | tr -d '\n' >> gpurun_out/traffic_${TAG}.jsonl; echo >> gpurun_out/traffic_${TAG}.jsonl
python tools/rocpd_summary.py gpurun_out/prof_${TAG}_wide --traffic k_decode_wide_all 64 13 ${TAG} "tools/bench_configs.py --config 13" | tr -d '\n' >> gpurun_out/traffic_${TAG}.jsonl; echo >> gpurun_out/traffic_${TAG}.jsonl
# summaries here, databases deleted: gpurun only brings back 64 MiB
for d in gpurun_out/prof_${TAG}_*; do
    case $d in *_decode) W=18;; *_decode_c4) W=32;; *_decode_c3) W=2;; *_wide) W=5;; *) W=20;; esac   # untimed launches of the profiled command (bench.py: 8 setup + 10 warm-up; bench_configs.py: --warmup 20)
    python tools/rocpd_summary.py $d $W > gpurun_out/$(basename $d | sed 's/^prof_//')_rocprofv3.txt 2>&1 || true
    find $d -name '*.db' -delete
done
echo "summaries done"
