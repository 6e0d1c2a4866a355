#!/bin/bash
# Round 5's one bounded kernel session, same box, alternating:
#   (a) fused 4:4:4 kernel: the shipped workgroup order against HVC_444_ORDER = stripe / run:98 / run:34 / split, on the shipped
#       library and on the traffic-only build (build/variants/libhvc_traffic.so, EXTRA=-DHVC_TRAFFIC_ONLY=1);
#   (b) K3: shipped against the traffic-only build and the HVC_ENCODE_MULHI=1 build (build/variants/libhvc_mulhi.so: 80 VALU
#       instructions per block fewer), then SQ counter passes of each (VALU / LDS busy against the kernel's cycles).
# usage: bash tools/gpu_kernel_session.sh TAG ROUNDS   -> gpurun_out/TAG_fused_order.txt, TAG_k3_ab.txt, TAG_k3_pmc.txt
TAG=${1:-r05c}; ROUNDS=${2:-3}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
T=$ROOT/build/variants/libhvc_traffic.so
M=$ROOT/build/variants/libhvc_mulhi.so
[ -f $T ] && [ -f $M ] || { echo "variant builds missing"; exit 1; }
mkdir -p gpurun_out
F=gpurun_out/${TAG}_fused_order.txt
echo "# fused 4:4:4 kernel (512 x 1080p), workgroup order (HVC_444_ORDER; - = shipped: runs of 64 tiles per XCD), $ROUNDS alternations, one box" > $F
for rep in $(seq $ROUNDS); do
  for lib in shipped traffic; do
    for o in - stripe run:98 run:34 split; do
      export HVC_JPEG_LIB=; [ $lib = traffic ] && export HVC_JPEG_LIB=$T
      [ -z "$HVC_JPEG_LIB" ] && unset HVC_JPEG_LIB
      if [ $o = - ]; then unset HVC_444_ORDER; else export HVC_444_ORDER=$o; fi
      echo -n "$lib order=$o  " >> $F
      python tools/bench_configs.py --config 7 --fused-only --steps 40 | grep -o '"verified": [a-z]*\|"fused_ms": [0-9.]*\|"frac_of_8TBps": [0-9.]*' | paste - - - >> $F
    done
  done
done
unset HVC_444_ORDER HVC_JPEG_LIB
echo fused done
K=gpurun_out/${TAG}_k3_ab.txt
echo "# K3 (256 x 4K 4:2:0): shipped / HVC_ENCODE_MULHI=1 build / traffic-only build, $ROUNDS alternations, one box" > $K
for rep in $(seq $ROUNDS); do
  for lib in shipped mulhi traffic; do
    unset HVC_JPEG_LIB; [ $lib = mulhi ] && export HVC_JPEG_LIB=$M; [ $lib = traffic ] && export HVC_JPEG_LIB=$T
    echo -n "K3 $lib  " >> $K
    python tools/bench_configs.py --config 5 --steps 40 | grep -o '"verified": [a-z]*\|"kernel_ms": [0-9.]*\|"frac_of_8TBps": [0-9.]*' | paste - - - >> $K
  done
done
unset HVC_JPEG_LIB
echo k3 ab done
P=gpurun_out/${TAG}_k3_pmc.txt
: > $P
for lib in shipped mulhi traffic; do
  unset HVC_JPEG_LIB; [ $lib = mulhi ] && export HVC_JPEG_LIB=$M; [ $lib = traffic ] && export HVC_JPEG_LIB=$T
  bash tools/gpu_pmc_cmd.sh ${TAG}_k3_${lib}_valu "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" tools/bench_configs.py --config 5 --steps 10 || true
  bash tools/gpu_pmc_cmd.sh ${TAG}_k3_${lib}_lds "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR" tools/bench_configs.py --config 5 --steps 10 || true
  echo "### K3 $lib" >> $P
  python tools/pmc_csv_summary.py --kernel k_encode gpurun_out/pmc_${TAG}_k3_${lib}_valu gpurun_out/pmc_${TAG}_k3_${lib}_lds >> $P 2>&1
done
unset HVC_JPEG_LIB
echo pmc done
