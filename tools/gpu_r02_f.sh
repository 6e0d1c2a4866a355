#!/bin/bash
# round-2 session: each kernel against the memory ceiling of its own access shape (traffic-only build), the new launch
# splitting at large batches, K3 with nt stores, full GPU suite
set -e
TAG=${1:-r02f}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/${TAG}_pytest.log 2>&1 || { tail -40 gpurun_out/${TAG}_pytest.log; exit 1; }
tail -3 gpurun_out/${TAG}_pytest.log
{
T=$ROOT/build/variants/libhvc_traffic.so
for rep in 1 2; do
echo "== K1 (bench.py, 1024 frames): shipped / traffic-only"
python bench.py --steps 40 --no-cpu-baseline | grep -o '"frac": [0-9.]*\|"kernel_ms": [0-9.]*' | paste - -
HVC_JPEG_LIB=$T python bench.py --steps 40 --no-cpu-baseline | grep -o '"frac": [0-9.]*\|"kernel_ms": [0-9.]*' | paste - -
echo "== K3 (config 5): shipped (nt) / plain stores / traffic-only"
python tools/bench_configs.py --config 5 | grep -o '"kernel_ms": [0-9.]*\|"frac_of_8TBps": [0-9.]*' | paste - -
HVC_JPEG_LIB=$ROOT/build/variants/libhvc_encplain.so python tools/bench_configs.py --config 5 | grep -o '"kernel_ms": [0-9.]*\|"frac_of_8TBps": [0-9.]*' | paste - -
HVC_JPEG_LIB=$T python tools/bench_configs.py --config 5 | grep -o '"kernel_ms": [0-9.]*\|"frac_of_8TBps": [0-9.]*' | paste - -
echo "== fused 4:4:4 (config 7): shipped / traffic-only"
python tools/bench_configs.py --config 7 | grep -o '"fused_ms": [0-9.]*\|"frac_of_8TBps": [0-9.]*' | paste - -
HVC_JPEG_LIB=$T python tools/bench_configs.py --config 7 | grep -o '"fused_ms": [0-9.]*\|"frac_of_8TBps": [0-9.]*' | paste - -
echo "== K2 (config 2)"
python tools/bench_configs.py --config 2 | grep -o '"kernel_ms": [0-9.]*\|"frac_of_8TBps": [0-9.]*' | paste - -
done
echo "== large batches with the launch splitting: 2048 / 4096 frames per call"
for f in 2048 4096; do python bench.py --frames $f --steps 20 --no-cpu-baseline | grep -o '"frames_per_launch": [0-9]*\|"frac": [0-9.]*\|"kernel_ms": [0-9.]*' | paste - - -; done
HVC_LAUNCH_BYTES=1e12 python bench.py --frames 4096 --steps 20 --no-cpu-baseline | grep -o '"frames_per_launch": [0-9]*\|"frac": [0-9.]*\|"kernel_ms": [0-9.]*' | paste - - -
} 2>/dev/null | tee gpurun_out/${TAG}_shapes.txt
