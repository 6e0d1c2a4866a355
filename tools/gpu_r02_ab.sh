#!/bin/bash
# round-2 A/B session: occupancy / store variants of K3 and the fused kernel, launch-size and footprint experiments,
# TLB / memory-path counters at three launch sizes
set -e
TAG=${1:-r02e}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
{
echo "== fused 4:4:4 (config 7): LDS pad -> workgroups per CU"
for pad in 0 16384 28672 36864 0; do echo -n "pad $pad: "; HVC_444_LDS_PAD=$pad python tools/bench_configs.py --config 7 | grep -o '"fused_ms": [0-9.]*\|"frac_of_8TBps": [0-9.]*' | paste - -; done
echo "== K3 (config 5): LDS pad / nt stores"
for pad in 0 8192 21504 0; do echo -n "pad $pad: "; HVC_ENC_LDS_PAD=$pad python tools/bench_configs.py --config 5 | grep -o '"kernel_ms": [0-9.]*\|"frac_of_8TBps": [0-9.]*' | paste - -; done
for v in encnt encw3nt encw4nt; do echo -n "$v: "; HVC_JPEG_LIB=$ROOT/build/variants/libhvc_$v.so python tools/bench_configs.py --config 5 | grep -o '"kernel_ms": [0-9.]*\|"frac_of_8TBps": [0-9.]*' | paste - -; done
echo "== config 2 over a 154 GB resident set in 1024-frame launches; config 4 in launches of 64 / 128 / 256 / 512 frames"
python bench.py --config 2 --frames 1024 --shard 16384 --steps 2 --warmup 1 --no-cpu-baseline | grep -o '"frac": [0-9.]*\|"kernel_ms": [0-9.]*\|"workload": "[^"]*"' | paste - - -
for f in 64 128 256 512; do python bench.py --config 4 --frames $f --steps 2 --warmup 1 --no-cpu-baseline | grep -o '"frames_per_launch": [0-9]*\|"frac": [0-9.]*\|"kernel_ms": [0-9.]*' | paste - - -; done
} 2>&1 | tee gpurun_out/${TAG}_ab.txt
cd /tmp && export TMPDIR=/tmp
for f in 1024 2048 4096; do
  for grp in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" "TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum GRBM_GUI_ACTIVE"; do
    d=$ROOT/gpurun_out/pmc_${TAG}_f$f/$(echo $grp | cut -d' ' -f1)
    mkdir -p $d
    timeout -k 10 200 rocprofv3 --pmc $grp -d $d -o pmc -- python3 $ROOT/bench.py --frames $f --steps 10 --warmup 2 --no-cpu-baseline > $d/log.txt 2>&1 || echo "pmc group failed: $grp"
  done
done
python3 $ROOT/tools/pmc_csv_summary.py $ROOT/gpurun_out/pmc_${TAG}_f* > $ROOT/gpurun_out/${TAG}_tlb_counters.txt 2>&1 || true
find $ROOT/gpurun_out/pmc_${TAG}_f* -name '*.db' -delete
cat $ROOT/gpurun_out/${TAG}_tlb_counters.txt
