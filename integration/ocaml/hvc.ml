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

(* int hvc_jpeg_get_yuv_frame(info, pixels, out, cap, out_len)               decoder.ml:403-420 *)
let jpeg_get_yuv_frame =
  foreign
    "hvc_jpeg_get_yuv_frame"
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

(* Plane.t (common/src/plane.ml:4-9) is a Base_bigstring = (char, int8_unsigned_elt, c_layout) Array1:
   its data pointer is passed zero-copy.  (Needs [Plane.plane : t -> Base_bigstring.t] exposed.) *)
let plane_ptr (p : Hardcaml_video_common.Plane.t) =
  bigarray_start array1 (Hardcaml_video_common.Plane.plane p)
;;

type coefs = (int, Bigarray.int16_signed_elt, Bigarray.c_layout) Bigarray.Array1.t

let coefs_ptr (c : coefs) = bigarray_start array1 c

(* Markers.Dqt.elements (zig-zag order) -> the uint16[64] the C side reads *)
let qtab_of_int_array (q : int array) =
  let a = CArray.make uint16_t 64 in
  Array.iteri (fun i v -> CArray.set a i (Unsigned.UInt16.of_int v)) q;
  a
;;
