/*
 * hvc_jpeg.h -- C ABI of libhvc_jpeg.so: the MI355X (gfx950) JPEG block-transform
 * path that drops in behind hardcamls/video-coding's `jpeg/model` Decoder /
 * Encoder API.
 *
 * The reference has no FFI layer; this header is the
 * boundary a maintainer binds at the seam `jpeg/model/src/decoder.ml:347-360`
 * (decode_block) and `jpeg/model/src/encoder.ml:195-205` (encode_block).  Each
 * entry point cites the reference code it replaces (paths relative to the
 * reference root).  INTEGRATION.md shows the OCaml ctypes binding.
 *
 * Conventions
 *  - plain C, no exceptions cross the boundary (every entry point is a function-try-block: std::bad_alloc comes
 *    back as HVC_E_OUT_OF_MEMORY, std::system_error as HVC_E_SYSTEM); every function returns
 *    HVC_OK (0) or a negative hvc_status; hvc_strerror() gives a static string.
 *    The OCaml wrapper maps a non-zero code to `raise_s [%message "hvc" ...]`,
 *    the model's own error style (decoder.ml:67, 92, 101, 136).
 *  - the caller owns every buffer it passes; the library never retains a
 *    caller pointer after the call returns (after hvc_synchronize for device
 *    pointers).  Device scratch and streams live inside hvc_ctx.
 *  - an hvc_ctx is bound to ONE GPU and is NOT thread-safe: one per host
 *    thread / GPU.  There is no CPU backend: hvc_create fails with
 *    HVC_E_NO_DEVICE when no gfx950 device is usable.
 *  - all results are bit-exact to the OCaml model: for every coefficient record (int16 coefficients, DC absolute;
 *    8- or 16-bit quantiser entries) and for every file the model decodes.  One thing the int16 RECORD cannot
 *    carry: the model's ints are 63-bit (decoder.ml:143 `dc = coefs.(0) + dc_pred` never wraps), so a malformed-
 *    but-decodable stream whose DC differences pile up to an absolute DC beyond +-32767 (17 blocks of +2047 in a
 *    row do it; no encoder writes that) still decodes there, to saturated blocks.  The entry points that RETURN
 *    records (hvc_jpeg_entropy_decode, hvc_jpeg_entropy_decode_gpu) refuse such a stream with HVC_E_RANGE instead
 *    of wrapping; the entry points that decode FILES TO PIXELS (hvc_jpeg_decode, hvc_jpeg_decode_yuv444, the
 *    hvc_jpeg_decode_batch family) carry those blocks' true DCs on a side list through an int64 fix-up and give the
 *    model's pixels (tests/test_host_entropy.py::test_dc_beyond_int16_is_refused_not_wrapped,
 *    tests/test_gpu_jpeg_api.py::test_dc_beyond_int16_decodes_like_the_model).  That includes DC categories of 17 to
 *    62 bits, which the model reads without complaint (decoder.ml:81-96 asks no question of the table) and then carries
 *    through its 63-bit arithmetic, wrap-around included: the int64 fix-up computes modulo 2^63 as OCaml does
 *    (tests/test_gpu_jpeg_api.py::test_dc_categories_up_to_62_bits).  From 63 bits on, mag' (decoder.ml:73-79) shifts by
 *    Sys.int_size or more, which OCaml leaves unspecified: there is no model result to match, HVC_E_BAD_JPEG.
 *    A component that comes out with ZERO width or height -- a sampling factor of zero other than the first component's,
 *    or a frame dimension of zero -- is the model's empty Plane.t: hvc_jpeg_read_header reports it (blocks_w or blocks_h
 *    of 0), the readers walk the MCUs with no block for it, the block stage skips it, and hvc_jpeg_decode succeeds exactly
 *    where Decoder.decode does; what the model then cannot do is make a Frame.t of such planes, and neither can
 *    hvc_jpeg_get_yuv_frame (see there).  A zero factor in the FIRST component raises in decode_seq (Division_by_zero,
 *    decoder.ml:377-382), in ALL components in init (Int.round_up to a multiple of 0): HVC_E_BAD_JPEG.
 *    An entropy-coded segment of at most 32 bits is decoded with
 *    Bitstream_reader.show's own length test (bitstream_reader.ml:31-33: a request for as many bits as the whole
 *    segment has raises), so such a file is refused or decoded exactly where the model refuses or decodes it
 *    (tests/test_host_entropy.py::test_segments_of_a_few_bytes_raise_where_the_model_does,
 *    ::test_tables_and_headers_the_model_raises_on).  The one input left without a counterpart: a scan with no marker
 *    behind it, on which the model's extract_entropy_coded_bits (decoder.ml:261-281) never returns; here the scan ends
 *    with the file.
 *
 * Data layouts
 *  - coefficients: int16, [plane][blocks_h][blocks_w][64], each block in
 *    ZIG-ZAG order exactly as the model's `coefs` array holds them after
 *    Huffman decoding (decoder.ml:118-140) but with the DC predictor already
 *    added (decoder.ml:143 is a sequential dependency and stays on the host).
 *  - quantiser tables: 64 x uint16 in zig-zag order = Markers.Dqt.elements
 *    (markers.ml:153-167), indexed like `qnt_tab.(i)` at decoder.ml:146.
 *  - pixel planes: row-major uint8, `stride` bytes per row = the model's
 *    Plane.t (common/src/plane.ml:4-17); a Base_bigstring's data pointer can be
 *    passed as is.
 */
#ifndef HVC_JPEG_H
#define HVC_JPEG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HVC_API __attribute__((visibility("default")))

typedef struct hvc_ctx hvc_ctx; /* opaque: device, stream, scratch, fix-up list */

typedef enum hvc_status {
    HVC_OK = 0,
    HVC_E_INVALID_ARG = -1,  /* null pointer, non-positive size, bad enum */
    HVC_E_NO_DEVICE = -2,    /* no usable gfx950 GPU / HIP runtime error at create */
    HVC_E_HIP = -3,          /* a HIP call failed; hvc_last_hip_error() has the code */
    HVC_E_ALIGNMENT = -4,    /* plane pointer/stride not 8-byte aligned, coefs not 16-byte aligned */
    HVC_E_RANGE = -5,        /* quantiser entry 0 on the ENCODER side (the model divides by it), encoder output outside int16, or (record-returning entry points) a
                                decoded absolute DC outside int16 */
    HVC_E_OUT_OF_MEMORY = -6,
    HVC_E_TOO_LARGE = -7,    /* plane geometry beyond the kernel's index range */
    HVC_E_BAD_JPEG = -8,     /* the model would raise: missing frame/scan/table, invalid Huffman code,
                                coefficient index out of range (decoder.ml:92, 101, 136, 228-243, 291) */
    HVC_E_UNSUPPORTED_MARKER = -9, /* "unsupported marker code" (decoder.ml:67) */
    HVC_E_SYSTEM = -10,      /* the system refused a resource the call needs: a host thread of the batch pipelines
                                could not be started (pids limit, RLIMIT_NPROC); the context stays usable */
    HVC_E_INTERNAL = -11,    /* an unexpected C++ exception was stopped at the boundary (never seen; reported, not thrown) */
    HVC_E_BUSY = -12         /* the slot of an asynchronous submission still holds one: hvc_wait(ctx, slot) first */
} hvc_status;

/* where the data pointers of a call live */
typedef enum hvc_mem {
    HVC_MEM_HOST = 0,   /* host memory: staged through device scratch; call blocks until done */
    HVC_MEM_DEVICE = 1  /* device memory of ctx's GPU: enqueued on ctx's stream, returns at once */
} hvc_mem;

HVC_API int hvc_create(hvc_ctx **out, int device);
HVC_API void hvc_destroy(hvc_ctx *ctx);
HVC_API const char *hvc_strerror(int code);
HVC_API int hvc_last_hip_error(const hvc_ctx *ctx);
/* "hvc_jpeg <version> (gfx950) kernels <id>": <id> = the first 12 hex digits of the SHA-256 of the kernel sources the library
 * was built from (csrc/Makefile KERNEL_ID) -- what ties a committed counter pass (profiles/traffic.json) to a build. */
HVC_API const char *hvc_version(void);

/* The host threads of the batch pipelines (hvc_jpeg_decode_batch*, hvc_jpeg_encode_batch*, the download threads of
 * the host-buffer entry points) live in the context: started by the first call that needs them (more when a later
 * call asks for more `threads`), reused by every call after that, joined by hvc_destroy.  A thread the system refuses
 * to start makes the call return HVC_E_SYSTEM; nothing is left running and the context stays usable.
 * hvc_host_threads reports how many the context holds and how many it has ever started (diagnostic).
 * hvc_host_threads_probe (no context, no GPU): starts `threads` pool threads the way a batch call would, runs an empty
 * task on each and joins them -- HVC_OK or HVC_E_SYSTEM; what tests/test_host_threads.py runs under RLIMIT_NPROC. */
HVC_API int hvc_host_threads(const hvc_ctx *ctx, int *alive, uint64_t *ever_started);
HVC_API int hvc_host_threads_probe(int threads);

/* Which CPUs those threads may run on.  No counterpart in the reference (the model is one
 * thread); it matters on a node with eight GPUs, where eight contexts each start `threads` workers: left alone they
 * wander over both sockets, away from the pinned rings they fill and from their GPU's PCIe root.
 *   cpulist  Linux list format, "0-15,32-47"; "auto" = the CPUs local to the context's GPU (the local_cpulist of its
 *            PCI function in sysfs: its NUMA node); NULL or "" = no restriction (the default).
 * CPUs outside the process's own affinity mask are dropped; HVC_E_INVALID_ARG for a malformed list or one that leaves
 * nothing.  The environment variable HVC_HOST_CPUS, if set, is applied by hvc_create in the same way (an unusable
 * value is ignored there).  hvc_get_host_cpus reports the list in force ("" = none) and the number of CPUs in it. */
HVC_API int hvc_set_host_cpus(hvc_ctx *ctx, const char *cpulist);
HVC_API int hvc_get_host_cpus(const hvc_ctx *ctx, char *out, size_t cap, int *n_cpus);

/* Use an existing HIP stream (hipStream_t passed as void*) so that a host runtime can order this
 * library's kernels with its own work; torch.cuda.current_stream().cuda_stream is such a handle.
 * NULL is a stream too -- HIP's default (null) stream, which is what PyTorch's default stream is --
 * NOT "back to the context's own stream": a fresh hvc_ctx runs on its own non-blocking stream, which
 * does not synchronise with the null stream; hvc_reset_stream returns to it. */
HVC_API int hvc_set_stream(hvc_ctx *ctx, void *hip_stream);
HVC_API int hvc_reset_stream(hvc_ctx *ctx);
HVC_API int hvc_synchronize(hvc_ctx *ctx);

/* HIP-event timer on ctx's stream (for benchmarks): begin, enqueue work, end. */
HVC_API int hvc_timer_begin(hvc_ctx *ctx);
HVC_API int hvc_timer_end(hvc_ctx *ctx, float *elapsed_ms);

/* Per-kernel timing: when enabled, every decode/encode call brackets its
 * dominant kernel (K1 / K3 / the fused 4:4:4 kernel -- not the fix-up or seam
 * kernels) with HIP events on ctx's stream (device-memory calls only); hvc_last_kernel_ms waits
 * for that kernel and returns its duration.  bench.py's roofline figure comes
 * from here. */
HVC_API int hvc_set_profiling(hvc_ctx *ctx, int enabled);
HVC_API int hvc_last_kernel_ms(hvc_ctx *ctx, float *elapsed_ms);
/* durations of the last n (<= 64) profiled calls, oldest first; does not
 * serialise the calls themselves (one event pair per call, ring of 64) */
HVC_API int hvc_kernel_ms_history(hvc_ctx *ctx, float *elapsed_ms, int n);

/* ------------------------------------------------------------------------- */
/* Decode side.
 *
 * hvc_dequant_idct_recon: for every 8x8 block of `n_planes` equally sized
 * component planes computes
 *     dequantize + inverse zig-zag   decoder.ml:142-149 (dc_pred = 0, DC absolute)
 *     Dct.Chen.inverse_8x8           dct.ml:11-107
 *     clip, +128, plane store        decoder.ml:213-224
 * i.e. the body of Decoder.decode_block (decoder.ml:347-360) minus the Huffman
 * part.  Block (bx,by) of plane p reads coefs + p*coef_plane_stride +
 * (by*blocks_w+bx)*64 and writes the 8x8 pixels at plane + p*plane_stride +
 * (by*8+j)*stride + bx*8+i.
 * coef_plane_stride is in int16 elements (0 = blocks_w*blocks_h*64);
 * plane_stride in bytes (0 = stride*blocks_h*8). */
HVC_API int hvc_dequant_idct_recon(hvc_ctx *ctx, const int16_t *coefs, size_t coef_plane_stride,
                                   const uint16_t *qtab, int blocks_w, int blocks_h, int n_planes,
                                   uint8_t *plane, size_t stride, size_t plane_stride, int where);

/* A frame batch in one launch: every frame has the same `n_comp` components
 * (Decoder.Component.t, decoder.ml:167-187; geometry of Decoder.init,
 * decoder.ml:304-345). */
typedef struct hvc_component {
    int blocks_w, blocks_h; /* decoded_width/8, decoded_height/8.  0 x n or n x 0: the model's empty plane (a sampling
                             * factor or frame dimension of zero, decoder.ml:304-345); the DECODING entry points skip such a
                             * component as decode_seq does, the encoding ones refuse it (the model's encoder has none) */
    int qtab;               /* index into qtabs[] */
    int reserved;
    size_t coef_offset;     /* int16 elements from the frame's coefficient record */
    size_t plane_offset;    /* bytes from the frame's pixel record */
    size_t stride;          /* bytes per pixel row (>= blocks_w*8, multiple of 8) */
} hvc_component;

/* coefs + f*coef_frame_stride + comp.coef_offset -> pixels + f*pixel_frame_stride
 * + comp.plane_offset, for f < n_frames.  qtabs: [n_qtabs][64] uint16 (host
 * memory always; tiny).  Same arithmetic as hvc_dequant_idct_recon. */
HVC_API int hvc_decode_frames(hvc_ctx *ctx, const int16_t *coefs, size_t coef_frame_stride,
                              const uint16_t *qtabs, int n_qtabs, const hvc_component *comps,
                              int n_comp, int n_frames, uint8_t *pixels,
                              size_t pixel_frame_stride, int where);

/* The step after the path, fused into it: 4:2:0 coefficient records in,
 * tight 4:4:4 frames out.  Per frame:
 *     the block stage of hvc_decode_frames on the three component planes
 *     Decoder.get_yuv_frame / crop        decoder.ml:403-420  (luma width x height, chroma /2)
 *     Planar_444.convert_from_420         tools/src/planar_444.ml:82-103, 122-131 (supersample_hv2)
 * Output record f = frames + f*frame_stride: planes Y, U, V, each width x height bytes, row-major,
 * stride = width, back to back.  The quarter-resolution chroma planes never reach memory.
 * comps: the 4:2:0 geometry of Decoder.init (blocks_w, blocks_h, qtab, coef_offset are used;
 * plane_offset / stride are ignored); n_comp must be 3, width and height even
 * (Yuv.assert_is_420, tools/src/yuv.ml:104-116) and inside the decoded planes. */
HVC_API int hvc_decode_frames_yuv444(hvc_ctx *ctx, const int16_t *coefs, size_t coef_frame_stride,
                                     const uint16_t *qtabs, int n_qtabs, const hvc_component *comps,
                                     int n_comp, int n_frames, int width, int height, uint8_t *frames,
                                     size_t frame_stride, int where);

/* Diagnostic: which implementation the decode entry points use.  0 = default (k_decode_packed, the
 * int16-pair block-per-lane kernel, with the int64 fix-up for blocks outside its proven range), 1 = the
 * unpacked int32 kernel k_decode_fast, 2 = the int64 kernel for every block, 3 = k_decode_q16, the
 * mapping BASELINE.json's north star describes (one block per quarter wavefront, coefficients staged in
 * LDS between the passes; 1.7x slower).  All four produce identical bytes; tests use
 * this to cross-check independent implementations at full batch sizes. */
HVC_API int hvc_set_decode_kernel(hvc_ctx *ctx, int which);

/* Number of blocks the last decode call on ctx routed through the wide
 * (64-bit) fix-up kernel (diagnostic; synchronises the stream). */
HVC_API int hvc_last_wide_blocks(hvc_ctx *ctx, uint64_t *count);

/* ------------------------------------------------------------------------- */
/* Encode side (mirror): level shift, Dct.Chen.forward_8x8, quantise, zig-zag
 *     Encoder.level_shifted_input_block  encoder.ml:81-90
 *     Dct.Chen.forward_8x8               dct.ml:109-196
 *     Encoder.quant / quant_and_scale    encoder.ml:98-108
 * Output coefficients in zig-zag order, DC absolute (the DC difference of
 * encoder.ml:138-140 stays on the host with the RLE/Huffman writer). */
HVC_API int hvc_fdct_quant(hvc_ctx *ctx, const uint8_t *plane, size_t stride, size_t plane_stride,
                           const uint16_t *qtab, int blocks_w, int blocks_h, int n_planes,
                           int16_t *coefs, size_t coef_plane_stride, int where);

HVC_API int hvc_encode_frames(hvc_ctx *ctx, const uint8_t *pixels, size_t pixel_frame_stride,
                              const uint16_t *qtabs, int n_qtabs, const hvc_component *comps,
                              int n_comp, int n_frames, int16_t *coefs, size_t coef_frame_stride,
                              int where);

/* hvc_encode_frames plus the debugging tail of Encoder.encode_block (encoder.ml:195-205) that an encoder created
 * with ~compute_reconstruction_error:true runs on every block -- from the block's quantised coefficients
 *     Encoder.dequant   encoder.ml:110-117   quant.(i) * table.(i) through Zigzag.inverse
 *     Encoder.idct      encoder.ml:94-96     Dct.Chen.inverse_8x8
 *     Encoder.recon     encoder.ml:119-125   recon = max 0 (min 255 (idct + 128)); error = abs (recon - input pixel)
 * recon and error (either may be NULL) are pixel records with the layout of `pixels` (same comps, strides and
 * frame stride; bytes outside the component planes are not touched): Block.Decoded.recon / .error of every
 * block at the block's place.  max 0 (min 255 (x + 128)) is the decoder's clip + level shift (decoder.ml:213-224),
 * so recon is also exactly what Decoder.decode will make of the file. */
HVC_API int hvc_encode_frames_recon(hvc_ctx *ctx, const uint8_t *pixels, size_t pixel_frame_stride,
                                    const uint16_t *qtabs, int n_qtabs, const hvc_component *comps, int n_comp,
                                    int n_frames, int16_t *coefs, size_t coef_frame_stride, uint8_t *recon,
                                    uint8_t *error, int where);

/* ------------------------------------------------------------------------- */
/* 4:2:0 -> 4:4:4 chroma upsample, tools/src/planar_444.ml:82-103
 * (supersample_hv2 over all rows, :122-131): src cw x ch -> dst 2cw x 2ch. */
HVC_API int hvc_upsample420(hvc_ctx *ctx, const uint8_t *src, int cw, int ch, size_t src_stride,
                            uint8_t *dst, size_t dst_stride, int n_planes, size_t src_plane_stride,
                            size_t dst_plane_stride, int where);

/* The rest of `oyuv convert` (tools/src/oconv.ml): the other resampling steps of Planar_444, plane by plane like
 * hvc_upsample420 (n_planes planes, src_plane_stride / dst_plane_stride bytes apart, 0 = tight; host or device memory):
 *   hvc_subsample420   Planar_444.subsample_hv2 over all rows (planar_444.ml:69-80, convert_to_420 :105-116):
 *                      src sw x sh -> dst (sw / 2) x (sh / 2), dst[c, r] = (a + b + c + d + 2) >> 2 of the 2 x 2 samples
 *   hvc_subsample422   Planar_444.subsample_h2 (planar_444.ml:18-23, convert_to_422 :35-44): src sw x sh -> dst (sw / 2) x sh
 *   hvc_upsample422    Planar_444.supersample_h2 (planar_444.ml:25-33, convert_from_422 :55-67): src cw x h -> dst 2cw x h,
 *                      dst[2c] = src[c], dst[2c + 1] = avg2 src[c] src[c + 1], the last column twice
 *   hvc_crop_planes    Yuv.crop (tools/src/yuv.ml:42-62) of one plane: dst[c, r] = src[clamp (c + x_pos), clamp (r + y_pos)]
 *                      -- a crop, an offset, and edge replication where the destination reaches past the source */
HVC_API int hvc_subsample420(hvc_ctx *ctx, const uint8_t *src, int sw, int sh, size_t src_stride, uint8_t *dst,
                             size_t dst_stride, int n_planes, size_t src_plane_stride, size_t dst_plane_stride, int where);
HVC_API int hvc_subsample422(hvc_ctx *ctx, const uint8_t *src, int sw, int sh, size_t src_stride, uint8_t *dst,
                             size_t dst_stride, int n_planes, size_t src_plane_stride, size_t dst_plane_stride, int where);
HVC_API int hvc_upsample422(hvc_ctx *ctx, const uint8_t *src, int cw, int h, size_t src_stride, uint8_t *dst,
                            size_t dst_stride, int n_planes, size_t src_plane_stride, size_t dst_plane_stride, int where);
HVC_API int hvc_crop_planes(hvc_ctx *ctx, const uint8_t *src, int sw, int sh, size_t src_stride, int x_pos, int y_pos,
                            uint8_t *dst, int dw, int dh, size_t dst_stride, int n_planes, size_t src_plane_stride,
                            size_t dst_plane_stride, int where);

/* Yuv_format.t (tools/src/yuv_format.ml): the planar formats by their usual number, the packed 4:2:2 ones by name */
enum { HVC_YUV_420 = 420, HVC_YUV_422 = 422, HVC_YUV_444 = 444, HVC_YUV_YUY2 = 1, HVC_YUV_UYVY = 2, HVC_YUV_YVYU = 3 };
/* bytes of one raw frame: Planar.create / Packed.create (yuv_format.ml:21-53: integer halves; packed = 2 * width * height) */
HVC_API int hvc_yuv_frame_bytes(int format, int width, int height, size_t *bytes);
/* Oconv.main's loop body (oconv.ml:111-133) for n_frames raw frames, back to back in `src` and `dst`:
 *   Oconv.input   the source format to a 4:4:4 frame (Packed_422.convert_to_planar, Planar_444.convert_from_420 / _422)
 *   Yuv.crop      ~x_pos:x_off ~y_pos:y_off into a dst_w x dst_h 4:4:4 frame (clamped source coordinates)
 *   Oconv.output  the destination format (Planar_444.convert_to_420 / _422, Packed_422.convert_from_planar)
 * HVC_E_INVALID_ARG where Yuv.assert_is_420 / _422 raise: an odd width for a subsampled format, an odd height for 4:2:0. */
HVC_API int hvc_yuv_convert(hvc_ctx *ctx, const uint8_t *src, int src_format, int src_w, int src_h, int x_off, int y_off,
                            uint8_t *dst, int dst_format, int dst_w, int dst_h, int n_frames, int where);

/* ------------------------------------------------------------------------- */
/* Host front end / back end around the block stage:
 * the callers and data formats either side of the hot path.  Host C++ only. */

typedef struct hvc_jpeg_component { /* Decoder.Component.t geometry, decoder.ml:167-187, 304-345 */
    int identifier, hscale, vscale;
    int decoded_width, decoded_height; /* padded plane: rounded to the MCU */
    int actual_width, actual_height;   /* cropped size of get_yuv_frame */
    int dc_table, ac_table;            /* Huffman table selectors of the scan */
} hvc_jpeg_component;

typedef struct hvc_jpeg_info {
    int width, height, n_comp;     /* Markers.Sof / Sos; components in scan order */
    int n_qtabs;
    hvc_jpeg_component comp[4];
    hvc_component layout[4];       /* tight frame record: planes / coefficient planes back to back */
    uint16_t qtabs[4][64];         /* tables layout[].qtab refers to (Markers.Dqt.elements order) */
    size_t coef_count;             /* int16 elements of one frame's coefficient record */
    size_t pixel_bytes;            /* bytes of one frame's padded pixel record */
    size_t ecs_offset;             /* byte offset of the entropy-coded segment */
} hvc_jpeg_info;

/* Decoder.Header.decode (decoder.ml:36-70) + the geometry of Decoder.init (:294-345). */
HVC_API int hvc_jpeg_read_header(const uint8_t *jpeg, size_t n, hvc_jpeg_info *info);
/* The Huffman + DC-prediction half of Decoder.decode in decode_seq order (decoder.ml:118-140, 143,
 * 261-281, 362-395) into one frame's coefficient record (host memory, info->coef_count int16). */
HVC_API int hvc_jpeg_entropy_decode(const uint8_t *jpeg, size_t n, const hvc_jpeg_info *info, int16_t *coefs);
/* An EXTENSION, off unless asked for: restart intervals (DRI + RSTn markers, ITU-T T.81 B.2.4.4, E.2.4).  The model parses
 * DRI and never looks at it again; its entropy-coded segment ends at the first RSTn as at any marker (decoder.ml:56-59,
 * 261-281), so it decodes the first interval and reads zeros from there on -- and so does every entry point of this library
 * by default (model parity).  hvc_jpeg_entropy_decode_restart is hvc_jpeg_entropy_decode with the markers honoured: every
 * interval's bytes behind its RSTn, DC predictors back to zero at each (SURVEY.md 8f next-1's named extension);
 * hvc_set_restart_markers(ctx, 1) makes the context's file-level entry points (hvc_jpeg_decode, hvc_jpeg_decode_yuv444, the
 * batch pipelines, hvc_jpeg_entropy_decode_gpu) do the same -- the host reader by walking from interval to interval, the GPU
 * reader by taking every interval as a stream of its own (an RSTn is a synchronisation point known in advance: a byte
 * boundary, the first block of an MCU, predictors at zero); files of one batch whose DRI differs from the first file's, or
 * whose markers are not the ones their DRI promises, are the host reader's.  A file without DRI decodes the same either way.
 * Which reader takes a call never changes its result, only its speed; the GPU reader's limits with the extension on are:
 * intervals-per-file x files of one call <= 65535 (hvc_jpeg_entropy_decode_gpu: *used_gpu = 0 past that), intervals per
 * file <= 4096 in the batch pipeline (the whole batch goes through the host-reader pipeline past that), and one DRI per
 * chunk (a chunk holding a file with another DRI is redone by the host reader; stats->entropy_ms_sum > 0 says some were).
 * A scan of a single interval (MCUs <= DRI) is read as the plain segment by both readers. */
HVC_API int hvc_jpeg_entropy_decode_restart(const uint8_t *jpeg, size_t n, const hvc_jpeg_info *info, int16_t *coefs);
HVC_API int hvc_set_restart_markers(hvc_ctx *ctx, int honour);
/* The same for TWO files on the calling thread, their symbols decoded in turn: a file is one stream and its symbols one
 * dependency chain (shift, table load, shift), two files are two chains the core overlaps -- 1.07x (Zen 5) to 1.3x (Golden
 * Cove) the files per second per thread on the bench's content (what the batch pipelines' workers do).  *status_a / *status_b receive
 * what hvc_jpeg_entropy_decode would have returned for each file; an error in one does not stop the other. */
HVC_API int hvc_jpeg_entropy_decode2(const uint8_t *jpeg_a, size_t n_a, const hvc_jpeg_info *info_a, int16_t *coefs_a,
                                     int *status_a, const uint8_t *jpeg_b, size_t n_b, const hvc_jpeg_info *info_b,
                                     int16_t *coefs_b, int *status_b);
/* Decoder.get_yuv_frame (decoder.ml:403-420): the crops of components 0, 1, 2 back to back (Frame.output order,
 * common/src/frame.ml:66-70) -- where Frame.of_planes (frame.ml:42-61) makes a frame of them: three components at least,
 * chroma planes of one size, and that size the luma plane's halved both ways, halved in width, or equal (C420 / C422 /
 * C444 by integer halves).  HVC_E_BAD_JPEG where of_planes raises (4:1:1, 4:4:0, one or two components, an empty plane
 * beside planes with samples); a fourth component is left out, as in the model.  *out_len: the bytes of the frame. */
HVC_API int hvc_jpeg_get_yuv_frame(const hvc_jpeg_info *info, const uint8_t *pixels, uint8_t *out, size_t cap,
                                   size_t *out_len);
/* Decoder.crop (decoder.ml:403-413) mapped over Decoder.get_decoded_planes (:399-401): EVERY component's crop back to
 * back in scan order, whatever the sampling -- for the files whose planes Frame.of_planes has no name for. */
HVC_API int hvc_jpeg_get_cropped_planes(const hvc_jpeg_info *info, const uint8_t *pixels, uint8_t *out, size_t cap,
                                        size_t *out_len);
/* Decoder.decode_a_frame minus the crop (decoder.ml:422-427): header, host entropy decode, GPU block
 * stage; `pixels` (host, info->pixel_bytes) receives the padded planes = get_decoded_planes. */
HVC_API int hvc_jpeg_decode(hvc_ctx *ctx, const uint8_t *jpeg, size_t n, hvc_jpeg_info *info, uint8_t *pixels,
                            size_t pixel_cap);

/* The same for a 4:2:0 file, straight to a tight 4:4:4 frame (hvc_decode_frames_yuv444):
 * decode_a_frame (decoder.ml:422-427) followed by Planar_444.of_420 (tools/src/planar_444.ml:133-137),
 * i.e. `model.exe decode frame` + `oyuv convert -format 420 ... 444`.  frame: 3 * width * height bytes.
 * HVC_E_INVALID_ARG for any other sampling or an odd frame size (Yuv.assert_is_420). */
HVC_API int hvc_jpeg_decode_yuv444(hvc_ctx *ctx, const uint8_t *jpeg, size_t n, hvc_jpeg_info *info,
                                   uint8_t *frame, size_t frame_cap);

/* BASELINE config 3: a batch of baseline JPEGs of identical geometry and tables.  `threads` host
 * threads run the entropy decode into pinned chunk buffers; each finished chunk goes to the GPU with
 * hipMemcpyAsync on a copy stream while the block-stage kernel of the previous chunk runs on the
 * compute stream and the host decodes the next one.  pixels: n padded pixel records
 * (pixel_frame_stride bytes apart; `where` says host or device memory). */
typedef struct hvc_batch_stats {
    double wall_ms, entropy_ms_sum, h2d_ms_sum, kernel_ms_sum, d2h_ms_sum; /* sums over chunks */
    int chunks, threads, frames_per_chunk;
    uint64_t coef_bytes;
    double host_prep_ms_sum; /* hvc_jpeg_encode_batch: plane padding into the pinned ring, summed over threads */
} hvc_batch_stats;
HVC_API int hvc_jpeg_decode_batch(hvc_ctx *ctx, const uint8_t *const *jpegs, const size_t *sizes, int n_frames,
                                  int threads, int frames_per_chunk, uint8_t *pixels, size_t pixel_frame_stride,
                                  int where, hvc_batch_stats *stats);
/* The same pipeline ending in the fused kernel: 4:2:0 files in, tight 4:4:4 frames out (3 * width *
 * height bytes each, frame_stride apart) -- hvc_jpeg_decode_yuv444 for a batch. */
HVC_API int hvc_jpeg_decode_batch_yuv444(hvc_ctx *ctx, const uint8_t *const *jpegs, const size_t *sizes,
                                         int n_frames, int threads, int frames_per_chunk, uint8_t *frames,
                                         size_t frame_stride, int where, hvc_batch_stats *stats);

/* Quant_tables.scale Quant_tables.luma/chroma quality (quant_tables.ml:139-147). */
HVC_API int hvc_quant_table(int chroma_table, int quality, uint16_t *out64);

/* The cram tests' verification harness (`oyuv compare`, tools/src/ocompare.ml:6-47):
 * max_difference, total_difference and square_error of two planes of n bytes (host memory).
 * mean_difference / mean_square_error / psnr (:30-59) are one float operation on top
 * (video-coding_amd/yuv.py).  Any of the three outputs may be NULL. */
HVC_API int hvc_compare_planes(const uint8_t *a, const uint8_t *b, size_t n, int *max_difference,
                               uint64_t *total_difference, uint64_t *square_error);
/* Encoder.Parameters.c420/c422/c444 + Encoder.create geometry (encoder.ml:287-349, 437-472): chroma is
 * 420, 422 or 444.  Fills the padded plane layout (zero padding, plane.ml:11-17) and the tables. */
HVC_API int hvc_jpeg_encoder_layout(int width, int height, int chroma, int quality, hvc_jpeg_info *info);
/* hvc_jpeg_decode_batch (or _yuv444 when yuv444 != 0) with the Huffman reader on the GPU as well
 * (hvc_jpeg_entropy_decode_gpu below): host threads only parse headers and unstuff the entropy-coded
 * segments into a pinned ring, ~1 MB per 1080p frame crosses PCIe instead of 6 MB of coefficients, and
 * the coefficient records are produced where the block stage reads them.  Every file is read with ITS OWN
 * Huffman tables (decoder.ml:238-259: the DHT segments of the file): a chunk whose files all carry the first
 * file's tables runs with them in LDS, any other chunk with per-frame tables in device memory -- a batch of files
 * with per-file optimised tables stays on the GPU.  Chunks holding a file that needs the host decoder (see
 * below) are redone by the host-decoder pipeline once the others are through: same output, same errors
 * (stats->entropy_ms_sum is the host decoding time spent on them, 0 when the GPU reader did everything).
 * frames_per_chunk < 1 = a quarter of the batch, between 64 and 256 (what measured best). */
HVC_API int hvc_jpeg_decode_batch_gpu(hvc_ctx *ctx, const uint8_t *const *jpegs, const size_t *sizes, int n_frames,
                                      int threads, int frames_per_chunk, uint8_t *pixels,
                                      size_t pixel_frame_stride, int where, int yuv444, hvc_batch_stats *stats);

/* Huffman DEcoding on the GPU (csrc/hvc_hdec.hip): the entropy-coded segments of n_frames files (one
 * geometry; Huffman tables per file) -> coefficient records exactly as
 * hvc_jpeg_entropy_decode writes them.  The segment is cut into 1024-bit subsequences, one lane each;
 * lanes start from guessed states, adopt their predecessor's exit state round after round until nothing
 * changes (Huffman streams re-synchronise), then decode once more writing coefficients; a prefix sum
 * resolves the DC predictor (decoder.ml:143).  Whatever the model raises on, a DC outside int16, tables
 * that are no prefix code, or a stream that ends early makes the call fall back to the host decoder, so the
 * result (and every error code) is the host decoder's.  *used_gpu (optional) tells which one ran.
 * info receives the header of jpegs[0].  The call returns when the records are complete. */
HVC_API int hvc_jpeg_entropy_decode_gpu(hvc_ctx *ctx, const uint8_t *const *jpegs, const size_t *sizes, int n_frames,
                                        int16_t *coefs, size_t coef_frame_stride, int where, hvc_jpeg_info *info,
                                        int *used_gpu);

/* The encoder's back end ON THE GPU (csrc/hvc_huff.hip): Encoder.rle + write_bits + Bitstream_writer
 * with byte stuffing and flush_with_1s (encoder.ml:127-193, 507-510; bitstream_writer.ml) for n_frames
 * coefficient records, as data-parallel passes -- every block's bit string depends only on its own
 * coefficients and on the DC of the block before it in scan order.  Frame f's entropy-coded segment
 * (what stands between the SOS header and EOI) is out[offsets[f] .. offsets[f+1]); offsets has
 * n_frames + 1 entries.  coefs, out and offsets live where `where` says; the call returns when the
 * result is complete (it has to read the status back).  HVC_E_RANGE: a value without a code in the
 * default tables (as hvc_jpeg_entropy_encode); HVC_E_INVALID_ARG: out_cap too small.
 * hvc_jpeg_header gives the bytes in front of the segment (SOI .. SOS); 0xFF 0xD9 (EOI) closes the file. */
HVC_API int hvc_huffman_encode_frames(hvc_ctx *ctx, const hvc_jpeg_info *info, const int16_t *coefs,
                                      size_t coef_frame_stride, int n_frames, uint8_t *out, size_t out_cap,
                                      uint64_t *offsets, int where);
HVC_API int hvc_jpeg_header(const hvc_jpeg_info *info, uint8_t *out, size_t cap, size_t *len);
/* Diagnostic: the code tables the encoder's back ends emit from -- Tables.Encoder.dc_table / ac_table of the default
 * specifications (tables.ml:27-45 canonical assignment, :504-545; Tables.Default = ITU-T T.81 Annex K.3), which the
 * reference's own test prints in full (jpeg/model/test/test_tables.ml:4-395 -> tests/golden/g8_code_tables.json).
 * table_set: 0 luma, 1 chroma.  codes[i] = (code << 5) | length, 0 = no code: i < 16 the DC category i, i = 16 +
 * ((run << 4) | size) the AC symbol.  where = HVC_MEM_HOST: the host coder's tables (ctx may be NULL); HVC_MEM_DEVICE:
 * the GPU coder's tables as they sit in ctx's device memory (uploaded if no call has needed them yet, then read back). */
HVC_API int hvc_huffman_code_tables(hvc_ctx *ctx, int table_set, int where, uint32_t *codes272);

/* HVC_OK when Encoder.encode_seq can walk this geometry; HVC_E_INVALID_ARG where the model raises
 * "[Plane.get] out of bounds" (encoder.ml:476-505 with plane.ml:43-50): the MCU grid of the luma
 * component reaches past a chroma plane for 4:2:0 / 4:2:2 frames of width (or height) 16k + 1.  The
 * encode entry points below return the same error for such frames. */
HVC_API int hvc_jpeg_encoder_check(const hvc_jpeg_info *info);
/* Encoder.write_headers + rle + write_bits + EOI over a coefficient record (encoder.ml:127-193,
 * 371-418, 476-510): byte-identical to Model.Encoder's output. */
HVC_API int hvc_jpeg_entropy_encode(const hvc_jpeg_info *info, const int16_t *coefs, uint8_t *out, size_t cap,
                                    size_t *out_len);
/* Encoder.encode_420/422/444 ~frame ~quality (encoder.ml:512-541): y/u/v are the tight planes of the
 * frame (Frame.create sizes); padding on the host, forward block stage AND Huffman coder on the GPU
 * (only the entropy-coded segment is downloaded), header + segment + EOI assembled into out. */
HVC_API int hvc_jpeg_encode(hvc_ctx *ctx, const uint8_t *y, const uint8_t *u, const uint8_t *v, int width,
                            int height, int chroma, int quality, uint8_t *out, size_t cap, size_t *out_len);

/* The same for a batch of equally sized frames (BASELINE config 5 end to end): frames[f] is one raw
 * planar frame as `model encode frame` reads it (Frame.input, common/src/frame.ml:72-76: the tight Y, U,
 * V planes back to back); jpegs[f] receives the file (capacity caps[f]; its length in sizes[f]), byte-
 * identical to Encoder.encode_420/422/444.  Host threads pad planes into a pinned ring, hipMemcpyAsync on a
 * copy stream, the forward block stage and the download of the coefficient records on the compute
 * stream, host threads RLE + Huffman -- three chunks in flight. */
HVC_API int hvc_jpeg_encode_batch(hvc_ctx *ctx, const uint8_t *const *frames, int n_frames, int width, int height,
                                  int chroma, int quality, int threads, int frames_per_chunk,
                                  uint8_t *const *jpegs, const size_t *caps, size_t *sizes,
                                  hvc_batch_stats *stats);
/* The same with the Huffman coder on the GPU as well (hvc_huffman_encode_frames): the coefficient records
 * never leave the device, only the packed entropy-coded segments come back (about 1/6 of the bytes), and
 * the host threads just pad planes and assemble header + segment + EOI.  Same files, byte for byte.
 * HVC_E_TOO_LARGE if a chunk's segments exceed twice the size of its raw frames (use the host-coder
 * variant for such content); stats->coef_bytes then counts the segment bytes downloaded. */
HVC_API int hvc_jpeg_encode_batch_gpu(hvc_ctx *ctx, const uint8_t *const *frames, int n_frames, int width,
                                      int height, int chroma, int quality, int threads, int frames_per_chunk,
                                      uint8_t *const *jpegs, const size_t *caps, size_t *sizes,
                                      hvc_batch_stats *stats);

/* K5 (no counterpart in the reference): what a benchmark or a pipeline produced, said in
 * 64 bits per record without bringing the records back.  For r < n_records
 *     sums[r] = SUM_i (byte_i + 1) * ((2 i + 1) * 0x9E3779B97F4A7C15)   mod 2^64,  i = byte index in record r
 * over data + r * record_stride, record_bytes bytes each.  data lives where `where` says (host data is uploaded
 * first); sums is host memory; the call returns when sums is complete.  The same 64 bits follow from three
 * lines of numpy on any machine (tests/helpers.py checksum_records), which is how the CPU suite pins the
 * benchmarks' outputs to the model. */
HVC_API int hvc_checksum_records(hvc_ctx *ctx, const void *data, size_t record_bytes, size_t record_stride,
                                 int n_records, uint64_t *sums, int where);

/* ------------------------------------------------------------------------- */
/* The asynchronous seam ("async batch API: submit(frame batch, stream slot) / wait(slot)"; BASELINE.json
 * north_star: "host-side Huffman decode feeds pinned coefficient buffers via hipMemcpyAsync on a side stream overlapped
 * with the IDCT kernel").  For the caller that keeps the reference's OWN sequential Huffman reader
 * (decoder.ml:118-140, untouched) and wants what hvc_jpeg_decode_batch gives the library's reader: while the GPU works
 * on batch k, the caller's reader fills batch k + 1.
 *
 * A context has HVC_SLOTS slots; a slot carries one batch in flight:
 *     submit:  host coefficient records --hipMemcpyAsync, copy stream--> device --block stage, ctx's stream--> device pixels
 *              --hipMemcpyAsync, download stream--> host pixel records          (decode; the encoder mirror runs the other way)
 * and returns at once.  Three streams, so slot k + 1's upload, slot k's kernel and slot k - 1's download overlap, and the
 * link carries both directions.  hvc_wait(ctx, slot) blocks until the slot's results are visible and frees the slot.
 * The host buffers of a submission must stay valid and unmodified (coefficients) / unread (pixels) until its hvc_wait
 * (or hvc_destroy): the copy engines read and write them on their own time, and memory freed under them is a GPU page
 * fault, not a status code.
 * They should be PINNED -- from hvc_host_alloc, or the caller's own page-aligned memory passed once through
 * hvc_host_register: from pageable memory hipMemcpyAsync stages through the runtime's bounce buffers and holds its
 * caller (the result is the same, the overlap is gone).
 * Like everything on a context the slot calls are not thread-safe among themselves; what the caller does meanwhile on
 * its own memory (filling the next slot's buffer, on any number of threads) is its business.
 * Errors: HVC_E_BUSY = the slot still holds a submission (hvc_wait first); everything hvc_decode_frames /
 * hvc_encode_frames answer to the same arguments; a HIP failure inside the slot's work is reported by ITS hvc_wait.
 * Measured on MI355X: 1080p 4:2:0 records from pinned slots of 64 frames, the caller's threads refilling
 * the next slot meanwhile: 18.4 - 18.7 Gpixel/s at 56 - 57 GB/s of upload (the link's rate), 17 Gpixel/s with the pixel
 * records coming back as well; the blocking HVC_MEM_HOST call on pageable memory: 15.8. */
enum { HVC_SLOTS = 4 };

/* Pinned host memory (hipHostMalloc): 4 KiB-aligned, usable as a Bigarray (Ctypes.bigarray_of_ptr) or any byte buffer. */
HVC_API int hvc_host_alloc(hvc_ctx *ctx, size_t bytes, void **out);
HVC_API int hvc_host_free(hvc_ctx *ctx, void *p);
/* Pins memory the caller already owns (hipHostRegister): [p, p + bytes) stays where it is and becomes a DMA source /
 * target; hvc_host_unregister before freeing it.  WHOLE PAGES ONLY -- p and bytes multiples of the page size
 * (posix_memalign / mmap / aligned_alloc; HVC_E_ALIGNMENT otherwise), pages that hold nothing but this buffer: the HIP
 * runtime finds registered memory by page, and a pageable buffer that shares a registered range's last page is copied
 * through that range's mapping until the GPU faults where it ends.  (A Bigarray.Array1 lives outside the OCaml heap and
 * never moves, but Bigarray.create's memory is malloc's: take hvc_host_alloc for those, as the patch's Hvc.pinned_* do.) */
HVC_API int hvc_host_register(hvc_ctx *ctx, void *p, size_t bytes);
HVC_API int hvc_host_unregister(hvc_ctx *ctx, void *p);

/* hvc_decode_frames (same arguments, same arithmetic: decoder.ml:142-149, 213-224; dct.ml:11-107) on host coefficient
 * records, asynchronously in `slot`.  pixels_where = HVC_MEM_HOST: the pixel records are downloaded into `pixels`
 * (only the bytes the kernels wrote: the component planes); HVC_MEM_DEVICE: `pixels` is device memory of ctx's GPU
 * and the block stage writes it directly (hvc_wait then says the kernel is done). */
HVC_API int hvc_decode_frames_submit(hvc_ctx *ctx, int slot, const int16_t *coefs, size_t coef_frame_stride,
                                     const uint16_t *qtabs, int n_qtabs, const hvc_component *comps, int n_comp,
                                     int n_frames, uint8_t *pixels, size_t pixel_frame_stride, int pixels_where);
/* The encoder mirror: hvc_encode_frames (encoder.ml:81-108; dct.ml:109-196) on host pixel records; the coefficient
 * records go to `coefs`, host (downloaded: the coefficient planes) or device as coefs_where says. */
HVC_API int hvc_encode_frames_submit(hvc_ctx *ctx, int slot, const uint8_t *pixels, size_t pixel_frame_stride,
                                     const uint16_t *qtabs, int n_qtabs, const hvc_component *comps, int n_comp,
                                     int n_frames, int16_t *coefs, size_t coef_frame_stride, int coefs_where);
/* Blocks until the slot's submission is complete (an idle slot: returns at once); the slot is free afterwards.
 * hvc_slot_query: the same question without waiting (*done = 1: hvc_wait will not block). */
HVC_API int hvc_wait(hvc_ctx *ctx, int slot);
HVC_API int hvc_slot_query(hvc_ctx *ctx, int slot, int *done);
/* What the slot's LAST completed submission cost, from HIP events on the three streams (valid after its hvc_wait):
 * upload, block stage (incl. the fix-up kernels), download; 0 for a stage it did not have. */
typedef struct hvc_slot_stats {
    double h2d_ms, kernel_ms, d2h_ms;
    uint64_t h2d_bytes, d2h_bytes;
} hvc_slot_stats;
HVC_API int hvc_slot_last_stats(hvc_ctx *ctx, int slot, hvc_slot_stats *stats);

/* Device memory helpers so that a binding needs no HIP of its own. */
HVC_API int hvc_device_alloc(hvc_ctx *ctx, size_t bytes, void **out);
HVC_API int hvc_device_free(hvc_ctx *ctx, void *p);
HVC_API int hvc_memcpy_h2d(hvc_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);
HVC_API int hvc_memcpy_d2h(hvc_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);

#ifdef __cplusplus
}
#endif
#endif /* HVC_JPEG_H */
