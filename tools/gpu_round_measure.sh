#!/bin/bash
# One GPU session that produces every number DESIGN.md quotes (run via gpurun from the repo root):
#   gpurun_out/m_*.json  bench lines, gpurun_out/prof_<tag>_*/ rocprofv3 databases -> gpurun_out/<tag>_*_rocprofv3.txt;
#   at the end of part a everything m_* becomes <tag>_lines.jsonl (every JSON line, tagged with its command's name),
#   <tag>_stderr.txt and <tag>_<name>.txt (tools/consolidate_session.py): what goes to profiles/
set -e
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
# PART=a: the bench lines and microbenchmarks; PART=b: the rocprofv3 profiles; default: both (a call has 20 minutes)
PART=${PART:-all}
if [ "$PART" != "b" ]; then
python bench.py > gpurun_out/m_bench.json 2> gpurun_out/m_bench.err
python bench.py --config 4 --no-cpu-baseline > gpurun_out/m_bench_c4.json 2> gpurun_out/m_bench_c4.err
HVC_BENCH_REHEARSAL=1 python bench.py --gpus 2 --steps 20 --no-cpu-baseline > gpurun_out/m_bench_rehearsal2.json 2> gpurun_out/m_bench_rehearsal2.err
python bench.py --frames 4096 --steps 20 --no-cpu-baseline > gpurun_out/m_bench_f4096.json 2> /dev/null
python tools/bench_sustained.py > gpurun_out/m_sustained.json 2> /dev/null
echo "bench done"
python tools/bench_configs.py --config 3 --frames 1024 --threads 16 > gpurun_out/m_c3.json 2> gpurun_out/m_c3.err
python tools/bench_configs.py --config 3 --frames 4096 --threads 16 > gpurun_out/m_c3_4096.json 2>> gpurun_out/m_c3.err
python tools/bench_configs.py --config 4 > gpurun_out/m_c4.json 2> gpurun_out/m_c4.err
python tools/bench_configs.py --config 5 > gpurun_out/m_c5.json 2> gpurun_out/m_c5.err
python tools/bench_configs.py --config 6 > gpurun_out/m_c6.json 2> gpurun_out/m_c6.err
python tools/bench_configs.py --config 7 > gpurun_out/m_c7.json 2> gpurun_out/m_c7.err
python tools/bench_configs.py --config 2 > gpurun_out/m_k2.json 2> gpurun_out/m_k2.err
python tools/bench_configs.py --config 10 > gpurun_out/m_sub420.json 2> gpurun_out/m_sub420.err
python tools/bench_configs.py --config 11 > gpurun_out/m_convert.json 2> gpurun_out/m_convert.err
python tools/bench_configs.py --config 8 > gpurun_out/m_c5_files.json 2> gpurun_out/m_c5_files.err
HVC_DECODE_KERNEL=q16 python bench.py --no-cpu-baseline > gpurun_out/m_bench_q16.json 2> gpurun_out/m_bench_q16.err
# the file-level pipelines with the entropy stages on the GPU, the GPU Huffman coder alone, one file at a time
python tools/bench_configs.py --config 3 --frames 1024 --threads 16 --gpu-entropy --chunk 64 > gpurun_out/m_c3_gpu_entropy.json 2> gpurun_out/m_c3g.err
python tools/bench_configs.py --config 3 --frames 4096 --threads 16 --gpu-entropy --chunk 256 > gpurun_out/m_c3_gpu_entropy_4096.json 2>> gpurun_out/m_c3g.err
python tools/bench_configs.py --config 3 --frames 4096 --threads 16 --gpu-entropy --chunk 256 --own-tables > gpurun_out/m_c3_gpu_entropy_4096_own_tables.json 2>> gpurun_out/m_c3g.err
python tools/bench_configs.py --config 3 --frames 4096 --threads 16 --gpu-entropy --chunk 256 --host-out > gpurun_out/m_c3_gpu_entropy_host_out.json 2>> gpurun_out/m_c3g.err
# ... files with a restart interval of a row of MCUs (and their own tables), honoured: the GPU reader, the host reader
python tools/bench_configs.py --config 3 --frames 4096 --threads 16 --gpu-entropy --chunk 256 --restart-interval 120 > gpurun_out/m_c3_gpu_entropy_4096_restart.json 2>> gpurun_out/m_c3g.err
python tools/bench_configs.py --config 3 --frames 1024 --threads 16 --restart-interval 120 > gpurun_out/m_c3_restart.json 2>> gpurun_out/m_c3.err
python tools/bench_configs.py --config 8 --gpu-entropy > gpurun_out/m_c5_files_gpu_entropy.json 2> gpurun_out/m_c5g.err
python tools/bench_configs.py --config 9 > gpurun_out/m_huffman_gpu.json 2> gpurun_out/m_huffman_gpu.err
python tools/bench_single.py > gpurun_out/m_single_file.jsonl 2> gpurun_out/m_single_file.err
python tools/bench_small_batches.py > gpurun_out/m_small_batches.jsonl 2> gpurun_out/m_small_batches.err
python tools/bench_photo.py > gpurun_out/m_photo.jsonl 2> gpurun_out/m_photo.err
HVC_BENCH_REHEARSAL=1 python bench.py --gpus 4 --steps 20 --no-cpu-baseline --sustain-seconds 0.5 > gpurun_out/m_bench_rehearsal4.json 2> gpurun_out/m_bench_rehearsal4.err
HVC_BENCH_REHEARSAL=1 python bench.py --gpus 4 --config 4 --steps 1 --warmup 0 --frames 16 --shard 64 --no-cpu-baseline --sustain-seconds 0 > gpurun_out/m_bench_rehearsal4_c4.json 2> gpurun_out/m_bench_rehearsal4_c4.err
echo "configs done"
# one hvc_jpeg_decode call on one 1080p file: the timeline of its launches
D=$ROOT/gpurun_out/prof_${TAG}_single; mkdir -p $D
(cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace -d $D -o trace -- python3 $ROOT/tools/trace_single_call.py --quality 3 > $D/log.txt 2>&1) || true
python3 tools/trace_single_call.py --timeline $D > gpurun_out/m_single_call_timeline.txt 2>&1 || true
rm -rf $D
# the GPU Huffman reader alone on one 256-file chunk, one stream: per-kernel durations that do not depend on what the
# other reader stream of the pipeline is doing (model's tables; every file with its own optimised tables)
for v in "" "--own-tables" "--restart-interval 120"; do
  n=reader_chunk$(echo "$v" | sed 's/--own-tables/_own_tables/; s/--restart-interval 120/_restart/')
  D=$ROOT/gpurun_out/prof_${TAG}_$n; mkdir -p $D
  (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats -d $D/trace -o trace -- python3 $ROOT/tools/bench_reader_chunk.py --files 256 --reps 4 $v > $D/trace.log 2>&1) || true
  { grep records_equal $D/trace.log; python tools/reader_chunk_ms.py $D/trace; } > gpurun_out/m_$n.txt 2>&1 || true
  rm -rf $D
done
echo "reader chunk done"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/mem_ubench3 tools/ubench/mem_ubench3.hip 2> /dev/null
/tmp/mem_ubench3 > gpurun_out/m_mem_ubench3.txt 2>&1 || true
# the ceilings by bytes per lane (the block kernels move 192 - 384 B per lane: their ceiling is not the 16-B-per-lane one),
# and the kernels' real / ideal load and store halves
for u in inflight_ubench shape_ubench k1_ubench k1_dma_ubench; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/$u tools/ubench/$u.hip 2> /dev/null && /tmp/$u > gpurun_out/m_$u.txt 2>&1 || true
done
bash tools/gpu_shape_ceilings.sh > gpurun_out/m_shape_ceilings.txt 2> /dev/null || true
echo "ceilings done"
python tools/consolidate_session.py gpurun_out m ${TAG}
fi
if [ "$PART" = "a" ]; then exit 0; fi
bash tools/gpu_profile.sh ${TAG}_decode
echo "decode profile done"
bash tools/gpu_profile.sh ${TAG}_decode_c4 --config 4 --steps 2 --warmup 1
echo "config-4 profile done"
bash tools/gpu_profile_cmd.sh ${TAG}_444 tools/bench_configs.py --config 7 --steps 10
echo "444 profile done"
bash tools/gpu_profile_cmd.sh ${TAG}_encode tools/bench_configs.py --config 5 --steps 10
echo "encode profile done"
bash tools/gpu_profile_cmd.sh ${TAG}_c3g tools/bench_configs.py --config 3 --frames 512 --threads 16 --gpu-entropy --chunk 256 --steps 2
echo "reader profile done"
bash tools/gpu_profile_cmd.sh ${TAG}_c3g_own tools/bench_configs.py --config 3 --frames 512 --threads 16 --gpu-entropy --chunk 256 --steps 2 --own-tables
echo "reader (own tables) profile done"
python tools/rocpd_summary.py gpurun_out/prof_${TAG}_decode --traffic k_decode_packed 1024 2 ${TAG} > gpurun_out/traffic_${TAG}.json 2> /dev/null || true
python tools/rocpd_summary.py gpurun_out/prof_${TAG}_decode_c4 --traffic k_decode_packed 128 4 ${TAG} > gpurun_out/traffic_${TAG}_c4.json 2> /dev/null || true
# summaries here, databases deleted: gpurun only brings back 64 MiB
for d in gpurun_out/prof_${TAG}_*; do
    case $d in *_decode) W=18;; *_decode_c4) W=32;; *) W=10;; esac   # untimed launches of the profiled command (bench.py: 8 setup + 10 warm-up)
    python tools/rocpd_summary.py $d $W > gpurun_out/$(basename $d | sed 's/^prof_//')_rocprofv3.txt 2>&1 || true
    find $d -name '*.db' -delete
done
echo "summaries done"
