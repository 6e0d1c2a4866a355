#!/bin/bash
# same-box A/B of one environment switch over a bench_configs.py command, alternating, N rounds:
#   bash tools/gpu_ab_env.sh TAG VAR "v0 v1 ..." ROUNDS -- --config 8 --threads 16 --gpu-entropy
set -e
TAG=$1; VAR=$2; VALS=$3; N=$4; shift 5
mkdir -p gpurun_out
OUT=gpurun_out/${TAG}_ab_${VAR}.txt
echo "# python tools/bench_configs.py $* with $VAR = $VALS, alternating" > $OUT
for round in $(seq 1 $N); do
  for v in $VALS; do
    echo "$VAR=$v: $(env $VAR=$v python tools/bench_configs.py "$@" 2>/dev/null | grep -o '"value": [0-9.]*\|"process_cpus_busy": [0-9.]*' | tr '\n' ' ')" | tee -a $OUT
  done
done
