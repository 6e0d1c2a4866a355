(* decoder_gpu.ml -- the patch of jpeg/model/src/decoder.ml described in INTEGRATION.md section 4.
   UNCOMPILED (no OCaml toolchain in the build image).  Written against the names of decoder.ml:
   Component.t (:167-187) gains one field
       coef_plane : Hvc.coefs   (decoded_width / 8 * decoded_height / 8 * 64 int16, zig-zag order, DC absolute)
   allocated in [init] (:324-341) next to [plane]. *)

(* phase 1, per block: decoder.ml:151-165 without dequantisation / IDCT / recon *)
let decode_block_host ~bits ~(component : Component.t) =
  clear_block component.coefs;
  huffman_decode
    ~bits
    ~coefs:component.coefs
    ~dc_tab:component.dc_tab
    ~ac_tab:component.ac_tab;
  let dc = component.coefs.(0) + component.dc_pred in
  (* decoder.ml:143: the predictor is a sequential dependency and stays here *)
  component.dc_pred <- dc;
  let blocks_w = component.decoded_width / 8 in
  let base = (((component.y / 8) * blocks_w) + (component.x / 8)) * 64 in
  component.coef_plane.{base} <- dc;
  for i = 1 to 63 do
    component.coef_plane.{base + i} <- component.coefs.(i)
  done
;;

(* replaces [decode] (decoder.ml:397): same block order as decode_seq (:374-395) for phase 1, then one
   FFI call per component plane for dequantise + inverse zig-zag + Chen-Wang IDCT + clip / +128 / store *)
let decode_gpu (hvc : Hvc.ctx) (decoder : t) =
  iterate_blocks decoder ~f:(fun component -> decode_block_host ~bits:decoder.bits ~component);
  Array.iter decoder.components ~f:(fun (c : Component.t) ->
      let qtab = Hvc.qtab_of_int_array c.quant_table in
      Hvc.check
        "hvc_dequant_idct_recon"
        (Hvc.dequant_idct_recon
           hvc
           (Hvc.coefs_ptr c.coef_plane)
           Unsigned.Size_t.zero
           (Ctypes.CArray.start qtab)
           (c.decoded_width / 8)
           (c.decoded_height / 8)
           1
           (Hvc.plane_ptr c.plane)
           (Unsigned.Size_t.of_int c.decoded_width)
           Unsigned.Size_t.zero
           Hvc.mem_host))
;;

(* get_decoded_planes / crop / get_yuv_frame / decode_a_frame (decoder.ml:399-427) are unchanged: the kernel wrote
   into the very Plane.t buffers [init] allocated.  For_testing.Sequenced.decode (:433-435) keeps the per-block CPU
   path for the RTL testbenches. *)

(* Alternative without any OCaml phase 1: the library's own front end + block stage, then the model's crop. *)
let decode_a_frame_gpu (hvc : Hvc.ctx) (jpeg : string) : Frame.t =
  let info = Ctypes.make Hvc.Jpeg_info.t in
  let n = Unsigned.Size_t.of_int (String.length jpeg) in
  Hvc.check "hvc_jpeg_read_header" (Hvc.jpeg_read_header jpeg n (Ctypes.addr info));
  let pixel_bytes = Unsigned.Size_t.to_int (Ctypes.getf info Hvc.Jpeg_info.pixel_bytes) in
  let pixels = Base_bigstring.create pixel_bytes in
  let pixels_ptr = Ctypes.bigarray_start Ctypes.array1 pixels in
  Hvc.check
    "hvc_jpeg_decode"
    (Hvc.jpeg_decode hvc jpeg n (Ctypes.addr info) pixels_ptr (Unsigned.Size_t.of_int pixel_bytes));
  (* planes are views into [pixels] at info.layout.(i).plane_offset (decoded_width x decoded_height each);
     crop to actual_width x actual_height as get_yuv_frame does (decoder.ml:403-420), or let the library do it:
     hvc_jpeg_get_yuv_frame fills a tight Y, U, V buffer in Frame.output order *)
  frame_of_padded_record info pixels
;;
