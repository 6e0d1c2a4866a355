#!/bin/bash
# round-3 A/B session 1: where the shape costs (tools/ubench/shape_ubench.hip), the fused 4:4:4 path's launch modes
# (HVC_444_MODE 0 = one kernel, 1 = luma through k_decode_packed then chroma, 2 = side by side), K3 held to 3 / 4 waves
set -e
TAG=${1:-r03b}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
{
echo "== parity of the fused path in modes 1 and 2"
for m in 1 2; do HVC_444_MODE=$m python -m pytest tests/test_gpu_yuv444.py -q -x -m gpu 2>&1 | tail -1; done
echo "== shape microbenchmark"
build/shape_ubench
echo "== fused 4:4:4 (config 7): launch modes, alternating"
for rep in 1 2 3; do
  for m in 0 1 2; do echo -n "mode $m: "; HVC_444_MODE=$m python tools/bench_configs.py --config 7 2>/dev/null | grep -o '"verified": [a-z]*\|"fused_ms": [0-9.]*\|"frac_of_8TBps": [0-9.]*' | paste - - -; done
done
for p in -1 1; do echo -n "mode 2, side stream priority $p: "; HVC_444_SIDE_PRIO=$p HVC_444_MODE=2 python tools/bench_configs.py --config 7 2>/dev/null | grep -o '"fused_ms": [0-9.]*\|"frac_of_8TBps": [0-9.]*' | paste - -; done
echo "== fused 4:4:4 at 4K 4:2:0 ... (config 7 is 1080p only); K1 alone for reference"
python bench.py --steps 40 --no-cpu-baseline --sustain-seconds 0 2>/dev/null | grep -o '"frac": [0-9.]*\|"kernel_ms": [0-9.]*' | paste - -
echo "== K3 (config 5): shipped / 4 waves / 3 waves, alternating"
for rep in 1 2 3; do
  for v in shipped encw4 encw3; do
    echo -n "$v: "
    if [ $v = shipped ]; then python tools/bench_configs.py --config 5 2>/dev/null | grep -o '"kernel_ms": [0-9.]*\|"frac_of_8TBps": [0-9.]*' | paste - -
    else HVC_JPEG_LIB=$ROOT/build/variants/libhvc_$v.so python tools/bench_configs.py --config 5 2>/dev/null | grep -o '"kernel_ms": [0-9.]*\|"frac_of_8TBps": [0-9.]*' | paste - -; fi
  done
done
} 2>&1 | tee gpurun_out/${TAG}_ab.txt
