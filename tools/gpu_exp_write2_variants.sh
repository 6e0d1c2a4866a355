set -e
for spec in "lds=video-coding_amd/libhvc_jpeg.so" "g256=video-coding_amd/libhvc_jpeg.so@HVC_WR_MODE=2" "wb8=build/variants/libhvc_wb8.so" "wb16=build/variants/libhvc_wb16.so"; do
  name=${spec%%=*}; lib=${spec#*=}; unset HVC_WR_MODE
  case $lib in *@*) export "${lib#*@}"; lib=${lib%%@*};; esac
  export HVC_JPEG_LIB=$PWD/$lib
  D=$PWD/gpurun_out/prof_r02aj_$name; mkdir -p $D
  (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats -d $D/trace -o trace -- python3 $OLDPWD/tools/bench_reader_chunk.py --files 256 --reps 4 > $D/trace.log 2>&1)
  echo "== $name: $(grep -o '"records_equal_host_reader": [a-z]*' $D/trace.log) $(python tools/reader_chunk_ms.py $D/trace | grep k_hd_write2)"
  rm -rf $D
done
