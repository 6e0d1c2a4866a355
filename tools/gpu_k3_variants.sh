#!/bin/bash
# K3 builds side by side, same box, alternating: bash tools/gpu_k3_variants.sh TAG ROUNDS name1 name2 ...
# (name = build/variants/libhvc_<name>.so; "shipped" = the library as built)
TAG=$1; ROUNDS=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
K=gpurun_out/${TAG}_k3_variants.txt
mkdir -p gpurun_out
echo "# K3 (256 x 4K 4:2:0), builds: $@; $ROUNDS alternations, one box" > $K
for rep in $(seq $ROUNDS); do
  for lib in "$@"; do
    unset HVC_JPEG_LIB; [ $lib != shipped ] && export HVC_JPEG_LIB=$ROOT/build/variants/libhvc_$lib.so
    echo -n "K3 $lib  " >> $K
    python tools/bench_configs.py --config 5 --steps 40 | grep -o '"verified": [a-z]*\|"kernel_ms": [0-9.]*\|"frac_of_8TBps": [0-9.]*' | paste - - - >> $K
  done
done
unset HVC_JPEG_LIB
cat $K
