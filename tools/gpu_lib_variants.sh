#!/bin/bash
# Builds of the library side by side on one command, same box, alternating:
#   bash tools/gpu_lib_variants.sh TAG ROUNDS k1|k3|444 name1 name2 ...
# (name = build/variants/libhvc_<name>.so; "shipped" = the library as built; every line carries the command's K5 verdict)
TAG=$1; ROUNDS=$2; WHAT=$3; shift 3
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
K=gpurun_out/${TAG}_${WHAT}_variants.txt
mkdir -p gpurun_out
case $WHAT in
  k1)  CMD="python bench.py --steps 40 --no-cpu-baseline --no-others --sustain-seconds 0"; PAT='"verified": [a-z]*\|"kernel_ms": [0-9.]*\|"frac": [0-9.]*';;
  k3)  CMD="python tools/bench_configs.py --config 5 --steps 40"; PAT='"verified": [a-z]*\|"kernel_ms": [0-9.]*\|"frac_of_8TBps": [0-9.]*';;
  444) CMD="python tools/bench_configs.py --config 7 --fused-only --steps 40"; PAT='"verified": [a-z]*\|"fused_ms": [0-9.]*\|"frac_of_8TBps": [0-9.]*';;
  *) echo "k1 | k3 | 444"; exit 1;;
esac
echo "# $WHAT: $CMD; builds: $@; $ROUNDS alternations, one box" > $K
for rep in $(seq $ROUNDS); do
  for lib in "$@"; do
    unset HVC_JPEG_LIB; [ $lib != shipped ] && export HVC_JPEG_LIB=$ROOT/build/variants/libhvc_$lib.so
    echo -n "$WHAT $lib  " >> $K
    $CMD 2> /dev/null | grep -o "$PAT" | head -3 | paste - - - >> $K
  done
done
unset HVC_JPEG_LIB
cat $K
