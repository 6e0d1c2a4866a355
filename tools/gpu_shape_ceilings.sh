#!/bin/bash
# Every block kernel against the memory ceiling of its own access shape: the shipped library, then the traffic-only build
# (build/variants/libhvc_traffic.so = make -C video-coding_amd/csrc OUT=../../build/variants/libhvc_traffic.so
# OBJDIR=../../build/obj_traffic EXTRA=-DHVC_TRAFFIC_ONLY=1: same loads / stores / LDS exchanges, no arithmetic),
# alternating, same box, same session.  Prints to stdout.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
T=$ROOT/build/variants/libhvc_traffic.so
[ -f $T ] || { echo "no traffic-only build"; exit 1; }
B="--steps 40 --no-cpu-baseline --no-others --sustain-seconds 0"
echo "# shipped library, then the traffic-only build (same loads / stores / LDS exchanges, no arithmetic), same box, same session"
for rep in 1 2; do
  echo -n "K1 shipped      "; python bench.py $B | grep -o '"frac": [0-9.]*\|"kernel_ms": [0-9.]*' | paste - -
  echo -n "K1 traffic-only "; HVC_JPEG_LIB=$T python bench.py $B | grep -o '"frac": [0-9.]*\|"kernel_ms": [0-9.]*' | paste - -
  echo -n "K3 shipped      "; python tools/bench_configs.py --config 5 | grep -o '"kernel_ms": [0-9.]*\|"frac_of_8TBps": [0-9.]*' | paste - -
  echo -n "K3 traffic-only "; HVC_JPEG_LIB=$T python tools/bench_configs.py --config 5 | grep -o '"kernel_ms": [0-9.]*\|"frac_of_8TBps": [0-9.]*' | paste - -
  echo -n "444 shipped      "; python tools/bench_configs.py --config 7 --fused-only | grep -o '"fused_ms": [0-9.]*\|"frac_of_8TBps": [0-9.]*' | paste - -
  echo -n "444 traffic-only "; HVC_JPEG_LIB=$T python tools/bench_configs.py --config 7 --fused-only | grep -o '"fused_ms": [0-9.]*\|"frac_of_8TBps": [0-9.]*' | paste - -
done
