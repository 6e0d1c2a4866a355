(* hvc.ml -- ctypes binding of include/hvc_jpeg.h (libhvc_jpeg.so, the MI355X block-transform path).
   UNCOMPILED: the build image has no OCaml toolchain; see README.md in this directory.
   Every [foreign] below names the C prototype it binds; argument order is the header's. *)
open Ctypes
open Foreign

type ctx = unit ptr

let ctx : ctx typ = ptr void

(* hvc_status -> the model's error style (decoder.ml:67, 92, 101, 136) *)
let strerror = foreign "hvc_strerror" (int @-> returning string)

let check what code =
  if code <> 0
  then Base.raise_s [%message "hvc" (what : string) (code : int) (strerror code : string)]
;;

let mem_host = 0
let mem_device = 1

(* int hvc_create(hvc_ctx **out, int device); void hvc_destroy(hvc_ctx *) *)
let create = foreign "hvc_create" (ptr ctx @-> int @-> returning int)
let destroy = foreign "hvc_destroy" (ctx @-> returning void)

let with_ctx ?(device = 0) f =
  let p = allocate ctx null in
  check "hvc_create" (create p device);
  Base.Exn.protect ~f:(fun () -> f !@p) ~finally:(fun () -> destroy !@p)
;;

(* typedef struct hvc_component { int blocks_w, blocks_h, qtab, reserved;
                                  size_t coef_offset, plane_offset, stride; } *)
module Component = struct
  type t

  let t : t structure typ = structure "hvc_component"
  let blocks_w = field t "blocks_w" int
  let blocks_h = field t "blocks_h" int
  let qtab = field t "qtab" int
  let reserved = field t "reserved" int
  let coef_offset = field t "coef_offset" size_t
  let plane_offset = field t "plane_offset" size_t
  let stride = field t "stride" size_t
  let () = seal t
end

(* typedef struct hvc_jpeg_component { int identifier, hscale, vscale, decoded_width, decoded_height,
                                       actual_width, actual_height, dc_table, ac_table; } *)
module Jpeg_component = struct
  type t

  let t : t structure typ = structure "hvc_jpeg_component"
  let identifier = field t "identifier" int
  let hscale = field t "hscale" int
  let vscale = field t "vscale" int
  let decoded_width = field t "decoded_width" int
  let decoded_height = field t "decoded_height" int
  let actual_width = field t "actual_width" int
  let actual_height = field t "actual_height" int
  let dc_table = field t "dc_table" int
  let ac_table = field t "ac_table" int
  let () = seal t
end

(* typedef struct hvc_jpeg_info { int width, height, n_comp, n_qtabs; hvc_jpeg_component comp[4];
     hvc_component layout[4]; uint16_t qtabs[4][64]; size_t coef_count, pixel_bytes, ecs_offset; } *)
module Jpeg_info = struct
  type t

  let t : t structure typ = structure "hvc_jpeg_info"
  let width = field t "width" int
  let height = field t "height" int
  let n_comp = field t "n_comp" int
  let n_qtabs = field t "n_qtabs" int
  let comp = field t "comp" (array 4 Jpeg_component.t)
  let layout = field t "layout" (array 4 Component.t)
  let qtabs = field t "qtabs" (array 256 uint16_t)
  let coef_count = field t "coef_count" size_t
  let pixel_bytes = field t "pixel_bytes" size_t
  let ecs_offset = field t "ecs_offset" size_t
  let () = seal t
end

(* int hvc_dequant_idct_recon(ctx, coefs, coef_plane_stride, qtab, blocks_w, blocks_h, n_planes,
                              plane, stride, plane_stride, where)            decoder.ml:142-149, 213-224; dct.ml:11-107 *)
let dequant_idct_recon =
  foreign
    "hvc_dequant_idct_recon"
    ~release_runtime_lock:true
    (ctx @-> ptr int16_t @-> size_t @-> ptr uint16_t @-> int @-> int @-> int @-> ptr char @-> size_t
    @-> size_t @-> int @-> returning int)
;;

(* int hvc_decode_frames(ctx, coefs, coef_frame_stride, qtabs, n_qtabs, comps, n_comp, n_frames,
                         pixels, pixel_frame_stride, where) *)
let decode_frames =
  foreign
    "hvc_decode_frames"
    ~release_runtime_lock:true
    (ctx @-> ptr int16_t @-> size_t @-> ptr uint16_t @-> int @-> ptr Component.t @-> int @-> int
    @-> ptr char @-> size_t @-> int @-> returning int)
;;

(* int hvc_decode_frames_yuv444(ctx, coefs, coef_frame_stride, qtabs, n_qtabs, comps, n_comp, n_frames,
                                width, height, frames, frame_stride, where)
   decoder.ml:403-420 + tools/src/planar_444.ml:82-131 fused into the block stage *)
let decode_frames_yuv444 =
  foreign
    "hvc_decode_frames_yuv444"
    ~release_runtime_lock:true
    (ctx @-> ptr int16_t @-> size_t @-> ptr uint16_t @-> int @-> ptr Component.t @-> int @-> int @-> int
    @-> int @-> ptr char @-> size_t @-> int @-> returning int)
;;

(* int hvc_fdct_quant(ctx, plane, stride, plane_stride, qtab, blocks_w, blocks_h, n_planes, coefs,
                      coef_plane_stride, where)                               encoder.ml:81-108; dct.ml:109-196 *)
let fdct_quant =
  foreign
    "hvc_fdct_quant"
    ~release_runtime_lock:true
    (ctx @-> ptr char @-> size_t @-> size_t @-> ptr uint16_t @-> int @-> int @-> int @-> ptr int16_t
    @-> size_t @-> int @-> returning int)
;;

(* whole-file entry points (host C++ front / back end + the GPU block stage) *)
let jpeg_read_header =
  foreign "hvc_jpeg_read_header" (string @-> size_t @-> ptr Jpeg_info.t @-> returning int)
;;

(* int hvc_jpeg_decode(ctx, jpeg, n, info, pixels, pixel_cap)                 decoder.ml:422-427 minus the crop *)
let jpeg_decode =
  foreign
    "hvc_jpeg_decode"
    ~release_runtime_lock:true
    (ctx @-> string @-> size_t @-> ptr Jpeg_info.t @-> ptr char @-> size_t @-> returning int)
;;

(* int hvc_jpeg_get_yuv_frame(info, pixels, out, cap, out_len)               decoder.ml:403-420; Frame.of_planes' raises = HVC_E_BAD_JPEG *)
let jpeg_get_yuv_frame =
  foreign
    "hvc_jpeg_get_yuv_frame"
    (ptr Jpeg_info.t @-> ptr char @-> ptr char @-> size_t @-> ptr size_t @-> returning int)
;;

(* int hvc_jpeg_get_cropped_planes(info, pixels, out, cap, out_len)          decoder.ml:399-413: crop of every plane *)
let jpeg_get_cropped_planes =
  foreign
    "hvc_jpeg_get_cropped_planes"
    (ptr Jpeg_info.t @-> ptr char @-> ptr char @-> size_t @-> ptr size_t @-> returning int)
;;

(* int hvc_jpeg_encode(ctx, y, u, v, width, height, chroma, quality, out, cap, out_len)   encoder.ml:512-541 *)
let jpeg_encode =
  foreign
    "hvc_jpeg_encode"
    ~release_runtime_lock:true
    (ctx @-> ptr char @-> ptr char @-> ptr char @-> int @-> int @-> int @-> int @-> ptr char @-> size_t
    @-> ptr size_t @-> returning int)
;;

(* int hvc_quant_table(chroma_table, quality, out64)                         quant_tables.ml:139-147 *)
let quant_table = foreign "hvc_quant_table" (int @-> int @-> ptr uint16_t @-> returning int)

(* int hvc_compare_planes(a, b, n, max_difference, total_difference, square_error)   tools/src/ocompare.ml:6-47 *)
let compare_planes =
  foreign
    "hvc_compare_planes"
    (ptr char @-> ptr char @-> size_t @-> ptr int @-> ptr uint64_t @-> ptr uint64_t @-> returning int)
;;

(* int hvc_encode_frames(ctx, pixels, pixel_frame_stride, qtabs, n_qtabs, comps, n_comp, n_frames, coefs,
                         coef_frame_stride, where)                            encoder.ml:81-108 for a frame batch *)
let encode_frames =
  foreign
    "hvc_encode_frames"
    ~release_runtime_lock:true
    (ctx @-> ptr char @-> size_t @-> ptr uint16_t @-> int @-> ptr Component.t @-> int @-> int
    @-> ptr int16_t @-> size_t @-> int @-> returning int)
;;

(* int hvc_encode_frames_recon(ctx, pixels, pixel_frame_stride, qtabs, n_qtabs, comps, n_comp, n_frames, coefs,
                               coef_frame_stride, recon, error, where)
   Encoder.encode_block with ~compute_reconstruction_error:true: encoder.ml:110-125, 195-205 *)
let encode_frames_recon =
  foreign
    "hvc_encode_frames_recon"
    ~release_runtime_lock:true
    (ctx @-> ptr char @-> size_t @-> ptr uint16_t @-> int @-> ptr Component.t @-> int @-> int
    @-> ptr int16_t @-> size_t @-> ptr char @-> ptr char @-> int @-> returning int)
;;

(* int hvc_jpeg_entropy_decode(jpeg, n, info, coefs)                         decoder.ml:118-140, 143, 362-395 *)
let jpeg_entropy_decode =
  foreign
    "hvc_jpeg_entropy_decode"
    (string @-> size_t @-> ptr Jpeg_info.t @-> ptr int16_t @-> returning int)
;;

(* int hvc_jpeg_decode_yuv444(ctx, jpeg, n, info, frame, frame_cap)          decode_a_frame + Planar_444.of_420 *)
let jpeg_decode_yuv444 =
  foreign
    "hvc_jpeg_decode_yuv444"
    ~release_runtime_lock:true
    (ctx @-> string @-> size_t @-> ptr Jpeg_info.t @-> ptr char @-> size_t @-> returning int)
;;

(* typedef struct hvc_batch_stats { double wall_ms, entropy_ms_sum, h2d_ms_sum, kernel_ms_sum, d2h_ms_sum;
     int chunks, threads, frames_per_chunk; uint64_t coef_bytes; double host_prep_ms_sum; } *)
module Batch_stats = struct
  type t

  let t : t structure typ = structure "hvc_batch_stats"
  let wall_ms = field t "wall_ms" double
  let entropy_ms_sum = field t "entropy_ms_sum" double
  let h2d_ms_sum = field t "h2d_ms_sum" double
  let kernel_ms_sum = field t "kernel_ms_sum" double
  let d2h_ms_sum = field t "d2h_ms_sum" double
  let chunks = field t "chunks" int
  let threads = field t "threads" int
  let frames_per_chunk = field t "frames_per_chunk" int
  let coef_bytes = field t "coef_bytes" uint64_t
  let host_prep_ms_sum = field t "host_prep_ms_sum" double
  let () = seal t
end

(* int hvc_jpeg_decode_batch(ctx, jpegs, sizes, n_frames, threads, frames_per_chunk, pixels, pixel_frame_stride,
                             where, stats)                                    BASELINE config 3 *)
let jpeg_decode_batch =
  foreign
    "hvc_jpeg_decode_batch"
    ~release_runtime_lock:true
    (ctx @-> ptr string @-> ptr size_t @-> int @-> int @-> int @-> ptr char @-> size_t @-> int
    @-> ptr Batch_stats.t @-> returning int)
;;

(* int hvc_jpeg_decode_batch_gpu(ctx, jpegs, sizes, n_frames, threads, frames_per_chunk, pixels,
                                 pixel_frame_stride, where, yuv444, stats) *)
let jpeg_decode_batch_gpu =
  foreign
    "hvc_jpeg_decode_batch_gpu"
    ~release_runtime_lock:true
    (ctx @-> ptr string @-> ptr size_t @-> int @-> int @-> int @-> ptr char @-> size_t @-> int @-> int
    @-> ptr Batch_stats.t @-> returning int)
;;

(* int hvc_checksum_records(ctx, data, record_bytes, record_stride, n_records, sums, where)   K5 *)
let checksum_records =
  foreign
    "hvc_checksum_records"
    ~release_runtime_lock:true
    (ctx @-> ptr void @-> size_t @-> size_t @-> int @-> ptr uint64_t @-> int @-> returning int)
;;

(* int hvc_set_stream(ctx, hip_stream); int hvc_reset_stream(ctx); int hvc_synchronize(ctx) *)
let set_stream = foreign "hvc_set_stream" (ctx @-> ptr void @-> returning int)
let reset_stream = foreign "hvc_reset_stream" (ctx @-> returning int)
let synchronize = foreign "hvc_synchronize" ~release_runtime_lock:true (ctx @-> returning int)

(* ---- the rest of include/hvc_jpeg.h, so that the binding names every entry point ---- *)

(* const char *hvc_version(void); int hvc_last_hip_error(ctx) *)
let version = foreign "hvc_version" (void @-> returning string)
let last_hip_error = foreign "hvc_last_hip_error" (ctx @-> returning int)

(* host threads of the batch pipelines: int hvc_host_threads(ctx, alive, ever_started); int hvc_host_threads_probe(threads);
   int hvc_set_host_cpus(ctx, cpulist); int hvc_get_host_cpus(ctx, out, cap, n_cpus) *)
let host_threads = foreign "hvc_host_threads" (ctx @-> ptr int @-> ptr uint64_t @-> returning int)
let host_threads_probe = foreign "hvc_host_threads_probe" (int @-> returning int)
let set_host_cpus = foreign "hvc_set_host_cpus" (ctx @-> string @-> returning int)
let get_host_cpus = foreign "hvc_get_host_cpus" (ctx @-> ptr char @-> size_t @-> ptr int @-> returning int)

(* timers, profiling, kernel choice: int hvc_timer_begin(ctx); int hvc_timer_end(ctx, ms); int hvc_set_profiling(ctx, enabled);
   int hvc_last_kernel_ms(ctx, ms); int hvc_kernel_ms_history(ctx, ms, n); int hvc_set_decode_kernel(ctx, which);
   int hvc_last_wide_blocks(ctx, count) *)
let timer_begin = foreign "hvc_timer_begin" (ctx @-> returning int)
let timer_end = foreign "hvc_timer_end" ~release_runtime_lock:true (ctx @-> ptr float @-> returning int)
let set_profiling = foreign "hvc_set_profiling" (ctx @-> int @-> returning int)
let last_kernel_ms = foreign "hvc_last_kernel_ms" (ctx @-> ptr float @-> returning int)
let kernel_ms_history = foreign "hvc_kernel_ms_history" (ctx @-> ptr float @-> int @-> returning int)
let set_decode_kernel = foreign "hvc_set_decode_kernel" (ctx @-> int @-> returning int)
let last_wide_blocks = foreign "hvc_last_wide_blocks" (ctx @-> ptr uint64_t @-> returning int)

(* int hvc_upsample420(ctx, src, cw, ch, src_stride, dst, dst_stride, n_planes, src_plane_stride, dst_plane_stride, where)
   tools/src/planar_444.ml:82-103 *)
let upsample420 =
  foreign
    "hvc_upsample420"
    ~release_runtime_lock:true
    (ctx @-> ptr char @-> int @-> int @-> size_t @-> ptr char @-> size_t @-> int @-> size_t @-> size_t @-> int
    @-> returning int)
;;

(* the rest of `oyuv convert` (tools/src/oconv.ml), plane by plane like hvc_upsample420:
   int hvc_subsample420(ctx, src, sw, sh, src_stride, dst, dst_stride, n_planes, src_plane_stride, dst_plane_stride, where)
   tools/src/planar_444.ml:69-80, 105-116 *)
let subsample420 =
  foreign
    "hvc_subsample420"
    ~release_runtime_lock:true
    (ctx @-> ptr char @-> int @-> int @-> size_t @-> ptr char @-> size_t @-> int @-> size_t @-> size_t @-> int
    @-> returning int)
;;

(* int hvc_subsample422(...)   tools/src/planar_444.ml:18-23, 35-44 *)
let subsample422 =
  foreign
    "hvc_subsample422"
    ~release_runtime_lock:true
    (ctx @-> ptr char @-> int @-> int @-> size_t @-> ptr char @-> size_t @-> int @-> size_t @-> size_t @-> int
    @-> returning int)
;;

(* int hvc_upsample422(...)    tools/src/planar_444.ml:25-33, 55-67 *)
let upsample422 =
  foreign
    "hvc_upsample422"
    ~release_runtime_lock:true
    (ctx @-> ptr char @-> int @-> int @-> size_t @-> ptr char @-> size_t @-> int @-> size_t @-> size_t @-> int
    @-> returning int)
;;

(* int hvc_crop_planes(ctx, src, sw, sh, src_stride, x_pos, y_pos, dst, dw, dh, dst_stride, n_planes, src_plane_stride,
   dst_plane_stride, where)   tools/src/yuv.ml:42-62 *)
let crop_planes =
  foreign
    "hvc_crop_planes"
    ~release_runtime_lock:true
    (ctx @-> ptr char @-> int @-> int @-> size_t @-> int @-> int @-> ptr char @-> int @-> int @-> size_t @-> int
    @-> size_t @-> size_t @-> int @-> returning int)
;;

(* int hvc_yuv_frame_bytes(format, width, height, bytes)   tools/src/yuv_format.ml:21-53 *)
let yuv_frame_bytes = foreign "hvc_yuv_frame_bytes" (int @-> int @-> int @-> ptr size_t @-> returning int)

(* int hvc_yuv_convert(ctx, src, src_format, src_w, src_h, x_off, y_off, dst, dst_format, dst_w, dst_h, n_frames, where)
   tools/src/oconv.ml:111-133 *)
let yuv_convert =
  foreign
    "hvc_yuv_convert"
    ~release_runtime_lock:true
    (ctx @-> ptr char @-> int @-> int @-> int @-> int @-> int @-> ptr char @-> int @-> int @-> int @-> int @-> int
    @-> returning int)
;;

(* an extension beyond the model (off by default): restart intervals honoured
   int hvc_jpeg_entropy_decode_restart(jpeg, n, info, coefs);  int hvc_set_restart_markers(ctx, honour) *)
let jpeg_entropy_decode_restart =
  foreign
    "hvc_jpeg_entropy_decode_restart"
    ~release_runtime_lock:true
    (string @-> size_t @-> ptr Jpeg_info.t @-> ptr int16_t @-> returning int)
;;

let set_restart_markers = foreign "hvc_set_restart_markers" (ctx @-> int @-> returning int)

(* int hvc_jpeg_entropy_decode2(jpeg_a, n_a, info_a, coefs_a, status_a, jpeg_b, n_b, info_b, coefs_b, status_b):
   two files decoded in turn on the calling thread *)
let jpeg_entropy_decode2 =
  foreign
    "hvc_jpeg_entropy_decode2"
    ~release_runtime_lock:true
    (string @-> size_t @-> ptr Jpeg_info.t @-> ptr int16_t @-> ptr int @-> string @-> size_t @-> ptr Jpeg_info.t
    @-> ptr int16_t @-> ptr int @-> returning int)
;;

(* int hvc_jpeg_entropy_decode_gpu(ctx, jpegs, sizes, n_frames, coefs, coef_frame_stride, where, info, used_gpu) *)
let jpeg_entropy_decode_gpu =
  foreign
    "hvc_jpeg_entropy_decode_gpu"
    ~release_runtime_lock:true
    (ctx @-> ptr string @-> ptr size_t @-> int @-> ptr int16_t @-> size_t @-> int @-> ptr Jpeg_info.t @-> ptr int
    @-> returning int)
;;

(* int hvc_jpeg_decode_batch_yuv444(ctx, jpegs, sizes, n_frames, threads, frames_per_chunk, frames, frame_stride, where, stats) *)
let jpeg_decode_batch_yuv444 =
  foreign
    "hvc_jpeg_decode_batch_yuv444"
    ~release_runtime_lock:true
    (ctx @-> ptr string @-> ptr size_t @-> int @-> int @-> int @-> ptr char @-> size_t @-> int
    @-> ptr Batch_stats.t @-> returning int)
;;

(* the encoder's side: int hvc_jpeg_encoder_layout(width, height, chroma, quality, info); int hvc_jpeg_encoder_check(info);
   int hvc_jpeg_header(info, out, cap, len); int hvc_jpeg_entropy_encode(info, coefs, out, cap, out_len)
   encoder.ml:347-349, 371-418, 437-472, 127-193 *)
let jpeg_encoder_layout =
  foreign "hvc_jpeg_encoder_layout" (int @-> int @-> int @-> int @-> ptr Jpeg_info.t @-> returning int)
;;

let jpeg_encoder_check = foreign "hvc_jpeg_encoder_check" (ptr Jpeg_info.t @-> returning int)

let jpeg_header =
  foreign "hvc_jpeg_header" (ptr Jpeg_info.t @-> ptr char @-> size_t @-> ptr size_t @-> returning int)
;;

let jpeg_entropy_encode =
  foreign
    "hvc_jpeg_entropy_encode"
    ~release_runtime_lock:true
    (ptr Jpeg_info.t @-> ptr int16_t @-> ptr char @-> size_t @-> ptr size_t @-> returning int)
;;

(* int hvc_huffman_encode_frames(ctx, info, coefs, coef_frame_stride, n_frames, out, out_cap, offsets, where):
   the GPU Huffman coder on device- or host-resident coefficient records *)
let huffman_encode_frames =
  foreign
    "hvc_huffman_encode_frames"
    ~release_runtime_lock:true
    (ctx @-> ptr Jpeg_info.t @-> ptr int16_t @-> size_t @-> int @-> ptr char @-> size_t @-> ptr uint64_t @-> int
    @-> returning int)
;;

(* int hvc_huffman_code_tables(ctx, table_set, where, codes272): the encoder back ends' code tables (diagnostic;
   Tables.Encoder.dc_table / ac_table, tables.ml:504-545) *)
let huffman_code_tables =
  foreign "hvc_huffman_code_tables" (ctx @-> int @-> int @-> ptr uint32_t @-> returning int)
;;

(* int hvc_jpeg_encode_batch(ctx, frames, n_frames, width, height, chroma, quality, threads, frames_per_chunk, jpegs, caps,
                             sizes, stats) and the same with the Huffman coder on the GPU      BASELINE config 5, files out *)
let jpeg_encode_batch =
  foreign
    "hvc_jpeg_encode_batch"
    ~release_runtime_lock:true
    (ctx @-> ptr (ptr char) @-> int @-> int @-> int @-> int @-> int @-> int @-> int @-> ptr (ptr char) @-> ptr size_t
    @-> ptr size_t @-> ptr Batch_stats.t @-> returning int)
;;

let jpeg_encode_batch_gpu =
  foreign
    "hvc_jpeg_encode_batch_gpu"
    ~release_runtime_lock:true
    (ctx @-> ptr (ptr char) @-> int @-> int @-> int @-> int @-> int @-> int @-> int @-> ptr (ptr char) @-> ptr size_t
    @-> ptr size_t @-> ptr Batch_stats.t @-> returning int)
;;

(* device memory for callers that keep records resident: int hvc_device_alloc(ctx, bytes, out); int hvc_device_free(ctx, p);
   int hvc_memcpy_h2d(ctx, dst, src, bytes); int hvc_memcpy_d2h(ctx, dst, src, bytes) *)
let device_alloc = foreign "hvc_device_alloc" (ctx @-> size_t @-> ptr (ptr void) @-> returning int)
let device_free = foreign "hvc_device_free" (ctx @-> ptr void @-> returning int)
let memcpy_h2d = foreign "hvc_memcpy_h2d" ~release_runtime_lock:true (ctx @-> ptr void @-> ptr void @-> size_t @-> returning int)
let memcpy_d2h = foreign "hvc_memcpy_d2h" ~release_runtime_lock:true (ctx @-> ptr void @-> ptr void @-> size_t @-> returning int)

(* ---- the asynchronous seam: pinned host memory and slots (include/hvc_jpeg.h, "The asynchronous seam") ----
   While the GPU works on the batch submitted to one slot, the caller -- this library's own sequential Huffman reader,
   decoder.ml:118-140 -- fills the pinned record of the next one.  [Decoder.decode_frames_gpu] is built on these. *)
let slots = 4 (* enum { HVC_SLOTS = 4 } *)

(* int hvc_host_alloc(ctx, bytes, out); int hvc_host_free(ctx, p): pinned memory (hipHostMalloc)
   int hvc_host_register(ctx, p, bytes); int hvc_host_unregister(ctx, p): the caller's own memory pinned in place -- whole
   pages only (posix_memalign'ed / mmap'ed memory; Bigarray.create's is malloc'ed: use [pinned_coefs] / [pinned_bytes]) *)
let host_alloc = foreign "hvc_host_alloc" (ctx @-> size_t @-> ptr (ptr void) @-> returning int)
let host_free = foreign "hvc_host_free" (ctx @-> ptr void @-> returning int)
let host_register = foreign "hvc_host_register" (ctx @-> ptr void @-> size_t @-> returning int)
let host_unregister = foreign "hvc_host_unregister" (ctx @-> ptr void @-> returning int)

(* int hvc_decode_frames_submit(ctx, slot, coefs, coef_frame_stride, qtabs, n_qtabs, comps, n_comp, n_frames, pixels,
                                pixel_frame_stride, pixels_where): hvc_decode_frames on host records, returns at once *)
let decode_frames_submit =
  foreign
    "hvc_decode_frames_submit"
    (ctx @-> int @-> ptr int16_t @-> size_t @-> ptr uint16_t @-> int @-> ptr Component.t @-> int @-> int
    @-> ptr char @-> size_t @-> int @-> returning int)
;;

(* int hvc_encode_frames_submit(ctx, slot, pixels, pixel_frame_stride, qtabs, n_qtabs, comps, n_comp, n_frames, coefs,
                                coef_frame_stride, coefs_where): the encoder mirror *)
let encode_frames_submit =
  foreign
    "hvc_encode_frames_submit"
    (ctx @-> int @-> ptr char @-> size_t @-> ptr uint16_t @-> int @-> ptr Component.t @-> int @-> int
    @-> ptr int16_t @-> size_t @-> int @-> returning int)
;;

(* int hvc_wait(ctx, slot): blocks (runtime lock released) until the slot's results are visible; int hvc_slot_query(ctx, slot, done) *)
let wait = foreign "hvc_wait" ~release_runtime_lock:true (ctx @-> int @-> returning int)
let slot_query = foreign "hvc_slot_query" (ctx @-> int @-> ptr int @-> returning int)

(* typedef struct hvc_slot_stats { double h2d_ms, kernel_ms, d2h_ms; uint64_t h2d_bytes, d2h_bytes; } *)
module Slot_stats = struct
  type t

  let t : t structure typ = structure "hvc_slot_stats"
  let h2d_ms = field t "h2d_ms" double
  let kernel_ms = field t "kernel_ms" double
  let d2h_ms = field t "d2h_ms" double
  let h2d_bytes = field t "h2d_bytes" uint64_t
  let d2h_bytes = field t "d2h_bytes" uint64_t
  let () = seal t
end

(* int hvc_slot_last_stats(ctx, slot, stats) *)
let slot_last_stats = foreign "hvc_slot_last_stats" (ctx @-> int @-> ptr Slot_stats.t @-> returning int)

(* Plane.t (common/src/plane.ml:4-9) is a Base_bigstring = (char, int8_unsigned_elt, c_layout) Array1:
   its data pointer is passed zero-copy.  (Needs [Plane.plane : t -> Base_bigstring.t] exposed.) *)
let plane_ptr (p : Hardcaml_video_common.Plane.t) =
  bigarray_start array1 (Hardcaml_video_common.Plane.plane p)
;;

type coefs = (int, Bigarray.int16_signed_elt, Bigarray.c_layout) Bigarray.Array1.t

let coefs_ptr (c : coefs) = bigarray_start array1 c

(* Pinned records as Bigarrays: [hvc_host_alloc]'s memory wrapped, not copied.  The Bigarray does not own it:
   [free_pinned] gives it back (after the last [wait] on a submission that used it). *)
let pinned_coefs (hvc : ctx) n : coefs =
  let p = allocate (ptr void) null in
  check "hvc_host_alloc" (host_alloc hvc (Unsigned.Size_t.of_int (2 * n)) p);
  bigarray_of_ptr array1 n Bigarray.int16_signed (from_voidp int16_t !@p)
;;

let pinned_bytes (hvc : ctx) n : Base_bigstring.t =
  let p = allocate (ptr void) null in
  check "hvc_host_alloc" (host_alloc hvc (Unsigned.Size_t.of_int n) p);
  bigarray_of_ptr array1 n Bigarray.char (from_voidp char !@p)
;;

let free_pinned_coefs (hvc : ctx) (c : coefs) =
  check "hvc_host_free" (host_free hvc (to_voidp (bigarray_start array1 c)))
;;

let free_pinned_bytes (hvc : ctx) (b : Base_bigstring.t) =
  check "hvc_host_free" (host_free hvc (to_voidp (bigarray_start array1 b)))
;;

(* Markers.Dqt.elements (zig-zag order) -> the uint16[64] the C side reads *)
let qtab_of_int_array (q : int array) =
  let a = CArray.make uint16_t 64 in
  Array.iteri (fun i v -> CArray.set a i (Unsigned.UInt16.of_int v)) q;
  a
;;
