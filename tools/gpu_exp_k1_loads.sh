#!/bin/bash
# experiment: what whole-line loads would buy K1 -- the shipped kernel, the traffic-only build of it (same loads /
# stores, no arithmetic) and the traffic-only build whose wavefront loads cover whole lines, alternating, same session
for rep in 1 2 3; do
  echo -n "K1 shipped                    "; python bench.py --steps 40 --no-cpu-baseline | grep -o '"frac": [0-9.]*\|"kernel_ms": [0-9.]*' | paste - -
  echo -n "K1 traffic-only               "; HVC_JPEG_LIB=$PWD/build/variants/libhvc_traffic.so python bench.py --steps 40 --no-cpu-baseline | grep -o '"frac": [0-9.]*\|"kernel_ms": [0-9.]*' | paste - -
  echo -n "K1 traffic-only, line loads   "; HVC_JPEG_LIB=$PWD/build/variants/libhvc_traffic2.so python bench.py --steps 40 --no-cpu-baseline | grep -o '"frac": [0-9.]*\|"kernel_ms": [0-9.]*' | paste - -
done 2>/dev/null
