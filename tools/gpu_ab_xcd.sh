#!/bin/bash
# same-box A/B of the workgroup -> tile mapping (csrc/hvc_kernels.hip xcd_work): HVC_XCD_RUN = 0 (as dispatched) against runs
# of R tiles per XCD, alternating, the block kernels one after the other.   bash tools/gpu_ab_xcd.sh "0 32 16 64" ROUNDS
RUNS=${1:-"0 32"}; ROUNDS=${2:-3}
B="--steps 40 --no-cpu-baseline --no-others --sustain-seconds 0"
for rep in $(seq $ROUNDS); do
  for v in $RUNS; do
    export HVC_XCD_RUN=$v
    echo -n "HVC_XCD_RUN=$v K1      "; python bench.py $B | grep -o '"frac": [0-9.]*\|"kernel_ms": [0-9.]*\|"verified": [a-z]*' | paste - - -
    echo -n "HVC_XCD_RUN=$v K1 c4   "; python tools/bench_configs.py --config 4 | grep -o '"kernel_ms": [0-9.]*\|"frac_of_8TBps": [0-9.]*\|"verified": [a-z]*' | paste - - -
    echo -n "HVC_XCD_RUN=$v K3      "; python tools/bench_configs.py --config 5 | grep -o '"kernel_ms": [0-9.]*\|"frac_of_8TBps": [0-9.]*\|"verified": [a-z]*' | paste - - -
    echo -n "HVC_XCD_RUN=$v fused   "; python tools/bench_configs.py --config 7 --fused-only | grep -o '"fused_ms": [0-9.]*\|"frac_of_8TBps": [0-9.]*\|"verified": [a-z]*' | paste - - -
    echo -n "HVC_XCD_RUN=$v K2      "; python tools/bench_configs.py --config 2 --steps 100 | grep -o '"kernel_ms": [0-9.]*\|"frac_of_8TBps": [0-9.]*\|"verified": [a-z]*' | paste - - -
  done
done
