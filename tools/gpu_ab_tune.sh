#!/bin/bash
# same-box A/Bs of the block kernels' launch parameters UNDER the XCD-aware workgroup order (they were last tuned under the
# plain order): workgroups per CU (an unused LDS reservation caps them), bytes per launch.   bash tools/gpu_ab_tune.sh ROUNDS
ROUNDS=${1:-2}
B="--steps 40 --no-cpu-baseline --no-others --sustain-seconds 0"
k1() { python bench.py $B "$@" | grep -o '"frac": [0-9.]*\|"kernel_ms": [0-9.]*\|"verified": [a-z]*' | paste - - -; }
k3() { python tools/bench_configs.py --config 5 | grep -o '"kernel_ms": [0-9.]*\|"frac_of_8TBps": [0-9.]*\|"verified": [a-z]*' | paste - - -; }
f4() { python tools/bench_configs.py --config 7 --fused-only | grep -o '"fused_ms": [0-9.]*\|"frac_of_8TBps": [0-9.]*\|"verified": [a-z]*' | paste - - -; }
for rep in $(seq $ROUNDS); do
  for pad in 0 53000 64000; do   # K1: 116 VGPRs = 4 workgroups per CU; 3; 2
    echo -n "K1 HVC_DEC_LDS_PAD=$pad   "; HVC_DEC_LDS_PAD=$pad k1
  done
  for pad in 0 8000 20000 40000; do   # K3: 32 KB of LDS per workgroup = 5 workgroups per CU; + 8 KB: 4; + 20 KB: 3; + 40 KB: 2
    echo -n "K3 HVC_ENC_LDS_PAD=$pad   "; HVC_ENC_LDS_PAD=$pad k3
  done
  for pad in 0 16000 32000; do
    echo -n "444 HVC_444_LDS_PAD=$pad   "; HVC_444_LDS_PAD=$pad f4
  done
  for lb in 5e9 10e9 20e9 40e9; do   # 4096 frames (38.5 GB) per call, cut into launches of at most this many bytes
    echo -n "K1 4096 frames HVC_LAUNCH_BYTES=$lb   "; HVC_LAUNCH_BYTES=$lb k1 --frames 4096 --steps 10
  done
done
