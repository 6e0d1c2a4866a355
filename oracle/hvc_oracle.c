/*
 * hvc_oracle.c -- CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
 *
 * A literal restatement, in plain C with 64-bit integers, of the software JPEG
 * model of hardcamls/video-coding (OCaml, `jpeg/model/src`, `common/src`,
 * `tools/src`).  OCaml `int` is 63-bit, so int64_t reproduces it exactly for
 * every value the model can reach from 16-bit coefficients (< 2^47).  Beyond
 * that -- a Huffman table may give a DC symbol up to 62 magnitude bits, and the
 * model reads them (decoder.ml:81-96) -- OCaml's ints wrap modulo 2^63 without
 * a word: sums and products are kept modulo 2^64 here (-fwrapv, oracle/Makefile)
 * and read as 63-bit numbers wherever the model looks at one (ocaml_int: the
 * operand of `asr`, the compares of clip); tests/test_oracle_golden.py holds
 * this against a big-integer restatement that wraps after every operation.
 *
 * Nothing in the product path (video-coding_amd/, libhvc_jpeg.so) includes,
 * links or calls this file.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py load liborc.so, and only as the checker.
 *
 * Parity pin: the oracle is checked against the reference's own golden
 * vectors (tests/golden/, see tests/test_oracle_golden.py):
 *   G1 test_chen_dct.ml:47-87, G2 test_decoder_accelerator.ml:209-376,
 *   G3 mini.jpg byte equality, G4 PSNR pins of the jpeg/test cram files,
 *   G5/G6 test_quant_tables.ml, G7 planar_444.ml:197-249,
 *   G8 test_encode_headers.ml:17-134.
 * The reference itself (OCaml) cannot be built in this image: no ocaml/dune.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * the reference root).
 */
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

typedef int64_t i64;

/* ------------------------------------------------------------------------- */
/* jpeg/model/src/zigzag.ml:3-69 (inverse), 71-137 (forward)                  */
/* inverse[zz] = raster ; forward[raster] = zz                                */
static const int ZZ_INV[64] = {
    0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,
    12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
    58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
static int ZZ_FWD[64];

/* jpeg/model/src/quant_tables.ml:3-69 (luma), 71-137 (chroma): the Annex-K
 * numbers in array order, used by the model as if in zig-zag order
 * (quant_tables.mli:5-6). */
static const int Q_LUMA[64] = {
    16, 11, 10, 16, 24,  40,  51,  61,  12, 12, 14, 19, 26,  58,  60,  55,
    14, 13, 16, 24, 40,  57,  69,  56,  14, 17, 22, 29, 51,  87,  80,  62,
    18, 22, 37, 56, 68,  109, 103, 77,  24, 35, 55, 64, 81,  104, 113, 92,
    49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
static const int Q_CHROMA[64] = {
    17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99,
    24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99,
    99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
    99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};

static void init_tables(void) {
    static int done = 0;
    if (done) return;
    for (int i = 0; i < 64; i++) ZZ_FWD[ZZ_INV[i]] = i;
    done = 1;
}

ORC_API const int *orc_zigzag_inverse(void) { init_tables(); return ZZ_INV; }
ORC_API const int *orc_zigzag_forward(void) { init_tables(); return ZZ_FWD; }
ORC_API const int *orc_quant_luma(void) { return Q_LUMA; }
ORC_API const int *orc_quant_chroma(void) { return Q_CHROMA; }

/* quant_tables.ml:139-147  clip / scale.  OCaml `/` truncates toward zero;
 * all operands are positive here. */
static i64 clip3(i64 x, i64 lo, i64 hi) { return x < lo ? lo : (x > hi ? hi : x); }

ORC_API void orc_quant_scale(const int *table, int q, int *out) {
    i64 qq = clip3(q, 1, 100);
    i64 s = qq < 50 ? 5000 / qq : 200 - 2 * qq;
    for (int i = 0; i < 64; i++) {
        i64 d = ((i64)table[i] * s + 50) / 100;
        out[i] = (int)clip3(d, 1, 255);
    }
}

/* ------------------------------------------------------------------------- */
/* jpeg/model/src/dct.ml:3-107   Chen-Wang integer IDCT                       */
#define W1 2841
#define W2 2676
#define W3 2408
#define W5 1609
#define W6 1108
#define W7 565

/* OCaml `asr` is an arithmetic shift (floor); `lsl` on a negative int is a
 * multiplication by 2^k, written as such here. */
/* An OCaml int is 63 bits wide: what is kept here modulo 2^64, read as the model would read it. */
static inline i64 ocaml_int(i64 x) { return (i64)((uint64_t)x << 1) >> 1; }
static inline i64 asr(i64 x, int k) { return ocaml_int(x) >> k; } /* gcc: arithmetic on signed */

/* dct.ml:11-54 */
static void idct_row(i64 *block, int row) {
    int off = row * 8;
    i64 x0 = block[off + 0] * 2048 + 128;
    i64 x1 = block[off + 4] * 2048;
    i64 x2 = block[off + 6];
    i64 x3 = block[off + 2];
    i64 x4 = block[off + 1];
    i64 x5 = block[off + 7];
    i64 x6 = block[off + 5];
    i64 x7 = block[off + 3];
    i64 x8;
    /* first stage */
    x8 = W7 * (x4 + x5);
    x4 = x8 + (W1 - W7) * x4;
    x5 = x8 - (W1 + W7) * x5;
    x8 = W3 * (x6 + x7);
    x6 = x8 - (W3 - W5) * x6;
    x7 = x8 - (W3 + W5) * x7;
    /* second stage */
    x8 = x0 + x1;
    x0 = x0 - x1;
    x1 = W6 * (x3 + x2);
    x2 = x1 - (W2 + W6) * x2;
    x3 = x1 + (W2 - W6) * x3;
    x1 = x4 + x6;
    x4 = x4 - x6;
    x6 = x5 + x7;
    x5 = x5 - x7;
    /* third stage */
    x7 = x8 + x3;
    x8 = x8 - x3;
    x3 = x0 + x2;
    x0 = x0 - x2;
    {
        i64 s = x4 + x5, d = x4 - x5;
        x2 = asr(181 * s + 128, 8);
        x4 = asr(181 * d + 128, 8);
    }
    /* fourth stage */
    block[off + 0] = asr(x7 + x1, 8);
    block[off + 1] = asr(x3 + x2, 8);
    block[off + 2] = asr(x0 + x4, 8);
    block[off + 3] = asr(x8 + x6, 8);
    block[off + 4] = asr(x8 - x6, 8);
    block[off + 5] = asr(x0 - x4, 8);
    block[off + 6] = asr(x3 - x2, 8);
    block[off + 7] = asr(x7 - x1, 8);
}

/* dct.ml:56-98 */
static void idct_col(i64 *block, int col) {
    i64 x0 = block[col + 8 * 0] * 256 + 8192;
    i64 x1 = block[col + 8 * 4] * 256;
    i64 x2 = block[col + 8 * 6];
    i64 x3 = block[col + 8 * 2];
    i64 x4 = block[col + 8 * 1];
    i64 x5 = block[col + 8 * 7];
    i64 x6 = block[col + 8 * 5];
    i64 x7 = block[col + 8 * 3];
    i64 x8;
    /* first stage */
    x8 = W7 * (x4 + x5) + 4;
    x4 = asr(x8 + (W1 - W7) * x4, 3);
    x5 = asr(x8 - (W1 + W7) * x5, 3);
    x8 = W3 * (x6 + x7) + 4;
    x6 = asr(x8 - (W3 - W5) * x6, 3);
    x7 = asr(x8 - (W3 + W5) * x7, 3);
    /* second stage */
    x8 = x0 + x1;
    x0 = x0 - x1;
    x1 = W6 * (x3 + x2) + 4;
    x2 = asr(x1 - (W2 + W6) * x2, 3);
    x3 = asr(x1 + (W2 - W6) * x3, 3);
    x1 = x4 + x6;
    x4 = x4 - x6;
    x6 = x5 + x7;
    x5 = x5 - x7;
    /* third stage */
    x7 = x8 + x3;
    x8 = x8 - x3;
    x3 = x0 + x2;
    x0 = x0 - x2;
    {
        i64 s = x4 + x5, d = x4 - x5;
        x2 = asr(181 * s + 128, 8);
        x4 = asr(181 * d + 128, 8);
    }
    /* fourth stage */
    block[col + 8 * 0] = asr(x7 + x1, 14);
    block[col + 8 * 1] = asr(x3 + x2, 14);
    block[col + 8 * 2] = asr(x0 + x4, 14);
    block[col + 8 * 3] = asr(x8 + x6, 14);
    block[col + 8 * 4] = asr(x8 - x6, 14);
    block[col + 8 * 5] = asr(x0 - x4, 14);
    block[col + 8 * 6] = asr(x3 - x2, 14);
    block[col + 8 * 7] = asr(x7 - x1, 14);
}

/* dct.ml:100-107 : rows 0..7 then columns 0..7, in place */
ORC_API void orc_idct_8x8(i64 *block) {
    for (int i = 0; i < 8; i++) idct_row(block, i);
    for (int i = 0; i < 8; i++) idct_col(block, i);
}

/* dct.ml:109-112 */
static inline i64 c4(i64 f, i64 g) { return asr(362 * (f + g), 9); }
static inline i64 c62(i64 f, i64 g) { return asr(196 * f + 473 * g, 9); }
static inline i64 c71(i64 f, i64 g) { return asr(100 * f + 502 * g, 9); }
static inline i64 c35(i64 f, i64 g) { return asr(426 * f + 284 * g, 9); }

/* dct.ml:114-149 (dct_col: base=col, step=8) and 151-187 (dct_row: base=row*8,
 * step=1): the two bodies are the same butterfly on a strided 8-vector. */
static void fdct_1d(i64 *block, int base, int step) {
    i64 *p = block + base;
    i64 a0 = p[0 * step] + p[7 * step];
    i64 c3 = p[0 * step] - p[7 * step];
    i64 a1 = p[1 * step] + p[6 * step];
    i64 c2 = p[1 * step] - p[6 * step];
    i64 a2 = p[2 * step] + p[5 * step];
    i64 c1 = p[2 * step] - p[5 * step];
    i64 a3 = p[3 * step] + p[4 * step];
    i64 c0 = p[3 * step] - p[4 * step];
    i64 b0 = a0 + a3;
    i64 b1 = a1 + a2;
    i64 b2 = a1 - a2;
    i64 b3 = a0 - a3;
    p[0 * step] = c4(b0, b1);
    p[4 * step] = c4(b0, -b1);
    p[2 * step] = c62(b2, b3);
    p[6 * step] = c62(b3, -b2);
    b0 = c4(c2, -c1);
    b1 = c4(c2, c1);
    a0 = c0 + b0;
    a1 = c0 - b0;
    a2 = c3 - b1;
    a3 = c3 + b1;
    p[1 * step] = c71(a0, a3);
    p[5 * step] = c35(a1, a2);
    p[3 * step] = c35(a2, -a1);
    p[7 * step] = c71(a3, -a0);
}

/* dct.ml:189-196 : columns first, then rows */
ORC_API void orc_fdct_8x8(i64 *block) {
    for (int i = 0; i < 8; i++) fdct_1d(block, i, 8);
    for (int i = 0; i < 8; i++) fdct_1d(block, i * 8, 1);
}

/* ------------------------------------------------------------------------- */
/* jpeg/model/src/decoder.ml:142-149                                          */
ORC_API i64 orc_dequantize_dc_pred_and_inverse_zigzag(const i64 *qnt_tab, i64 dc_pred,
                                                      const i64 *coefs, i64 *dequant) {
    i64 dc = ocaml_int(coefs[0] + dc_pred);
    dequant[0] = dc * qnt_tab[0];
    for (int i = 1; i < 64; i++) dequant[ZZ_INV[i]] = coefs[i] * qnt_tab[i];
    return dc;
}

/* decoder.ml:213 */
static inline i64 clip_pix(i64 x) { x = ocaml_int(x); return x < -128 ? -128 : (x > 127 ? 127 : x); }

/* decoder.ml:215-224 : clip (mutating idct), +128, store at plane (x+i, y+j).
 * Plane is row-major u8 with `stride` bytes per row (common/src/plane.ml:45-61).
 * Returns -1 if the store would be out of bounds (the model raises). */
static int recon_block(i64 *idct, i64 *recon, uint8_t *plane, int pw, int ph, size_t stride,
                       int x, int y) {
    for (int j = 0; j < 8; j++) {
        for (int i = 0; i < 8; i++) {
            int k = i + j * 8;
            idct[k] = clip_pix(idct[k]);
            recon[k] = idct[k] + 128;
            if (x + i < 0 || x + i >= pw || y + j < 0 || y + j >= ph) return -1;
            plane[(size_t)(y + j) * stride + (size_t)(x + i)] = (uint8_t)recon[k];
        }
    }
    return 0;
}

/* The block stage of decoder.ml:347-360 without the Huffman part, on the
 * C-ABI's batch layout (include/hvc_jpeg.h): coefs[n_planes][bh][bw][64]
 * int16, zig-zag order, DC absolute (predictor already added, so dc_pred = 0
 * here); qtab 64 x u16 in zig-zag order; planes of (bw*8) x (bh*8) pixels,
 * `stride` bytes per row, `plane_stride` bytes between planes.  One block at a
 * time, the model's scalar loop structure.  This is the function bench.py's
 * cpu_baseline times. */
ORC_API int orc_dequant_idct_recon(const int16_t *coefs, const uint16_t *qtab, int bw, int bh,
                                   int n_planes, uint8_t *plane, size_t stride,
                                   size_t plane_stride) {
    init_tables();
    i64 q[64], c[64], dq[64], id[64], rc[64];
    for (int i = 0; i < 64; i++) q[i] = qtab[i];
    for (int p = 0; p < n_planes; p++) {
        uint8_t *pl = plane + (size_t)p * plane_stride;
        for (int by = 0; by < bh; by++) {
            for (int bx = 0; bx < bw; bx++) {
                const int16_t *src = coefs + (((size_t)p * bh + by) * bw + bx) * 64;
                for (int i = 0; i < 64; i++) c[i] = src[i];
                orc_dequantize_dc_pred_and_inverse_zigzag(q, 0, c, dq);
                memcpy(id, dq, sizeof id); /* Array.blito, decoder.ml:357 */
                orc_idct_8x8(id);
                if (recon_block(id, rc, pl, bw * 8, bh * 8, stride, bx * 8, by * 8)) return -1;
            }
        }
    }
    return 0;
}

/* Single-block form exposing the model's Component.Summary fields
 * (decoder.ml:189-203): dequant (raster), idct (after clip), recon. */
ORC_API void orc_decode_block_summary(const i64 *coefs_zz_diff, const i64 *qtab, i64 dc_pred,
                                      i64 *dequant, i64 *idct, i64 *recon, i64 *dc_out) {
    init_tables();
    uint8_t tmp[64];
    *dc_out = orc_dequantize_dc_pred_and_inverse_zigzag(qtab, dc_pred, coefs_zz_diff, dequant);
    memcpy(idct, dequant, 64 * sizeof(i64));
    orc_idct_8x8(idct);
    recon_block(idct, recon, tmp, 8, 8, 8, 0, 0);
}

/* ------------------------------------------------------------------------- */
/* jpeg/model/src/encoder.ml:81-108                                           */

/* encoder.ml:98-101 : C `/` truncates toward zero like OCaml's. */
static inline i64 quant_and_scale(i64 fdct, i64 qnt) {
    return fdct < 0 ? (fdct - qnt * 2) / (qnt * 4) : (fdct + qnt * 2) / (qnt * 4);
}

/* encoder.ml:81-90 + 92 + 103-108 for one block at (x_pos,y_pos) of a plane.
 * quant[] comes out in zig-zag order. */
/* (a quantiser entry of zero: quant_and_scale raises Division_by_zero in the model; callers test with quant_table_divides) */
static int quant_table_divides(const i64 *table) {
    for (int i = 0; i < 64; i++) if (table[i] == 0) return 0;
    return 1;
}
static void encode_block_stage(const uint8_t *plane, size_t stride, int x_pos, int y_pos,
                               const i64 *table, i64 *fdct, i64 *quant) {
    for (int y = 0; y < 8; y++)
        for (int x = 0; x < 8; x++) {
            int k = y * 8 + x;
            i64 p = plane[(size_t)(y + y_pos) * stride + (size_t)(x + x_pos)];
            fdct[k] = p - 128;
        }
    orc_fdct_8x8(fdct);
    for (int i = 0; i < 64; i++) quant[ZZ_FWD[i]] = quant_and_scale(fdct[i], table[ZZ_FWD[i]]);
}

/* Batch form on the C-ABI layout (mirror of orc_dequant_idct_recon). */
ORC_API int orc_fdct_quant(const uint8_t *plane, size_t stride, size_t plane_stride,
                           const uint16_t *qtab, int bw, int bh, int n_planes, int16_t *coefs) {
    init_tables();
    i64 q[64], fd[64], qu[64];
    for (int i = 0; i < 64; i++) q[i] = qtab[i];
    if (!quant_table_divides(q)) return -2; /* Division_by_zero */
    for (int p = 0; p < n_planes; p++) {
        const uint8_t *pl = plane + (size_t)p * plane_stride;
        for (int by = 0; by < bh; by++)
            for (int bx = 0; bx < bw; bx++) {
                int16_t *dst = coefs + (((size_t)p * bh + by) * bw + bx) * 64;
                encode_block_stage(pl, stride, bx * 8, by * 8, q, fd, qu);
                for (int i = 0; i < 64; i++) {
                    if (qu[i] < -32768 || qu[i] > 32767) return -1;
                    dst[i] = (int16_t)qu[i];
                }
            }
    }
    return 0;
}

/* encoder.ml:110-125 + 94-96 + 195-205: the debugging tail of Encoder.encode_block when the encoder was
 * created with ~compute_reconstruction_error:true -- from the block's quantised coefficients
 *   dequant :110-117  c = quant.(i) * table.(i) -> dequant / idct .(Zigzag.inverse.(i))
 *   idct    :94-96    Dct.Chen.inverse_8x8
 *   recon   :119-125  recon.(i) = max 0 (min 255 (idct.(i) + 128)); error.(i) = abs (recon.(i) - input_pixels.(i))
 * on the C-ABI batch layout of orc_fdct_quant; recon / error planes have the layout of `plane`. */
ORC_API int orc_encode_recon(const uint8_t *plane, size_t stride, size_t plane_stride, const uint16_t *qtab, int bw,
                             int bh, int n_planes, int16_t *coefs, uint8_t *recon, uint8_t *error) {
    init_tables();
    i64 q[64], fd[64], qu[64], idct[64];
    for (int i = 0; i < 64; i++) q[i] = qtab[i];
    if (!quant_table_divides(q)) return -2; /* Division_by_zero */
    for (int p = 0; p < n_planes; p++) {
        const uint8_t *pl = plane + (size_t)p * plane_stride;
        for (int by = 0; by < bh; by++)
            for (int bx = 0; bx < bw; bx++) {
                int16_t *dst = coefs + (((size_t)p * bh + by) * bw + bx) * 64;
                encode_block_stage(pl, stride, bx * 8, by * 8, q, fd, qu);
                for (int i = 0; i < 64; i++) {
                    if (qu[i] < -32768 || qu[i] > 32767) return -1;
                    dst[i] = (int16_t)qu[i];
                }
                for (int i = 0; i < 64; i++) idct[ZZ_INV[i]] = qu[i] * q[i];       /* dequant */
                orc_idct_8x8(idct);                                                /* idct */
                for (int y = 0; y < 8; y++)
                    for (int x = 0; x < 8; x++) {                                  /* recon */
                        const size_t at = (size_t)p * plane_stride + (size_t)(by * 8 + y) * stride + (size_t)(bx * 8 + x);
                        i64 r = idct[y * 8 + x] + 128;
                        r = r > 255 ? 255 : r;
                        r = r < 0 ? 0 : r;
                        i64 e = r - (i64)plane[at];
                        recon[at] = (uint8_t)r;
                        error[at] = (uint8_t)(e < 0 ? -e : e);
                    }
            }
    }
    return 0;
}

/* ------------------------------------------------------------------------- */
/* common/src/bitstream_reader.ml:7-57                                        */
typedef struct {
    const uint8_t *buf;
    i64 len;            /* bytes */
    i64 length_in_bits; /* :16 */
    i64 bit_pos;
} Bits;

static void bits_create(Bits *b, const uint8_t *buf, i64 len) {
    b->buf = buf; b->len = len; b->length_in_bits = len * 8; b->bit_pos = 0;
}
/* :19-22 get_byte: out-of-range reads give '\000' */
static int bits_get_byte(const Bits *b, i64 byte_no) {
    return (byte_no >= 0 && byte_no < b->len) ? b->buf[byte_no] : 0;
}
/* :24-29 */
static int bits_get_bit(const Bits *b, i64 pos) {
    i64 byte_no = pos >> 3;
    int bit_no = 7 - (int)(pos & 7);
    return (bits_get_byte(b, byte_no) >> bit_no) & 1;
}
#define ORC_E_BITS_OOB (-2) /* "Bitstream_reader out of bounds" :32 */
/* :31-38 show ; returns <0 on the raise */
static i64 bits_show(const Bits *b, int n) {
    if (n >= b->length_in_bits) return ORC_E_BITS_OOB;
    i64 v = 0;
    for (int i = 0; i < n; i++) v = (v << 1) | bits_get_bit(b, b->bit_pos + i);
    return v;
}
static void bits_advance(Bits *b, i64 n) { b->bit_pos += n; }         /* :40 */
static i64 bits_get(Bits *b, int n) {                                  /* :42-46 */
    i64 v = bits_show(b, n);
    bits_advance(b, n);
    return v;
}
static void bits_align_to_byte(Bits *b) {                              /* :51-54 */
    i64 nb = b->bit_pos & 7;
    if (nb) bits_advance(b, 8 - nb);
}

/* ------------------------------------------------------------------------- */
/* common/src/bitstream_writer.ml:3-49                                        */
typedef struct {
    uint64_t word_buffer;
    int word_bits;
    uint8_t *buffer;
    size_t n, cap;
    i64 bytes_written;
} Writer;

static void w_init(Writer *w) { memset(w, 0, sizeof *w); }
static void w_add_char(Writer *w, int c) {
    if (w->n == w->cap) {
        w->cap = w->cap ? w->cap * 2 : 16384;
        w->buffer = (uint8_t *)realloc(w->buffer, w->cap);
    }
    w->buffer[w->n++] = (uint8_t)c;
}
/* :19-30 flush */
static void w_flush(Writer *w, int stuffing) {
    while (w->word_bits >= 8) {
        int d = (int)((w->word_buffer >> (w->word_bits - 8)) & 0xff);
        w_add_char(w, d);
        w->bytes_written++;
        w->word_bits -= 8;
        if (stuffing && d == 0xff) {
            w_add_char(w, 0);
            w->bytes_written++;
        }
    }
}
/* :32-40 put_bits (bits <= 16).  The OCaml word_buffer is never cleared; only
 * its low word_bits bits are ever read, so a wrapping uint64 is equivalent. */
static void w_put_bits(Writer *w, int stuffing, i64 value, int bits) {
    if (bits == 0) return;
    w->word_buffer = (w->word_buffer << bits) | ((uint64_t)value & ((1ull << bits) - 1));
    w->word_bits += bits;
    w_flush(w, stuffing);
}
static i64 w_bits_written(const Writer *w) { return w->bytes_written * 8 + w->word_bits; } /* :43 */
/* :45-49 */
static void w_flush_with_1s(Writer *w, int stuffing) {
    while (w_bits_written(w) & 7) w_put_bits(w, stuffing, 1, 1);
}

/* ------------------------------------------------------------------------- */
/* jpeg/model/src/tables.ml                                                   */
typedef struct { int length; int bits; int data; } Code; /* data: dc cat, or (run<<4)|size */

typedef struct {
    int lengths[16];
    int values[256];
    int nvalues;
} HuffSpec;

/* tables.ml:27-45 create_code_table: canonical code assignment */
static int create_code_table(const HuffSpec *s, Code *out) {
    int n = 0, code = 0, data_pos = 0;
    for (int lp = 0; lp < 16; lp++) {
        if (s->lengths[lp] == 0) {
            code <<= 1;
        } else {
            for (int i = 0; i < s->lengths[lp]; i++) {
                out[n].length = lp + 1;
                out[n].bits = code + i;
                out[n].data = s->values[data_pos + i];
                n++;
            }
            code = (code + s->lengths[lp]) << 1;
            data_pos += s->lengths[lp];
        }
    }
    return n;
}

/* tables.ml:478-502 Lut.create: (1 << max_bits) entries, None = length 0 */
typedef struct { int max_bits; int *len; int *data; } Lut;

/* 0, or -1 where the model raises: a specification whose codes do not fit the code space (more codes of some length
 * than canonical assignment has room for) makes Lut.create index past its array -- "index out of bounds" */
static int lut_create(Lut *l, const Code *codes, int n) {
    int max_bits = 0;
    for (int i = 0; i < n; i++) if (codes[i].length > max_bits) max_bits = codes[i].length;
    l->max_bits = max_bits;
    l->len = (int *)calloc((size_t)1 << max_bits, sizeof(int));
    l->data = (int *)calloc((size_t)1 << max_bits, sizeof(int));
    for (int i = 0; i < n; i++) {
        int null_bits = max_bits - codes[i].length;
        int first = codes[i].bits << null_bits;
        int count = 1 << null_bits;
        if (first < 0 || (i64)first + count > ((i64)1 << max_bits)) return -1;
        for (int k = first; k < first + count; k++) {
            l->len[k] = codes[i].length;
            l->data[k] = codes[i].data;
        }
    }
    return 0;
}
static void lut_free(Lut *l) { free(l->len); free(l->data); l->len = l->data = NULL; }

/* tables.ml:54-476 Default: ITU-T T.81 Annex K.3 typical Huffman tables. */
static const HuffSpec DC_LUMA = {{0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0},
                                 {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11}, 12};
static const HuffSpec DC_CHROMA = {{0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0},
                                   {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11}, 12};
static const HuffSpec AC_LUMA = {
    {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d},
    {0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61,
     0x07, 0x22, 0x71, 0x14, 0x32, 0x81, 0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52,
     0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25,
     0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45,
     0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64,
     0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83,
     0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99,
     0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6,
     0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3,
     0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8,
     0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa},
    162};
static const HuffSpec AC_CHROMA = {
    {0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77},
    {0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61,
     0x71, 0x13, 0x22, 0x32, 0x81, 0x08, 0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33,
     0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18,
     0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44,
     0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63,
     0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a,
     0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97,
     0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4,
     0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca,
     0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7,
     0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa},
    162};

/* tables.ml:504-545 Encoder.dc_table / ac_table.  dc: codes sorted by
 * category, indexed by size.  ac: sorted by (run,size), grouped by run, a fake
 * size-0 entry prepended to groups lacking one -> indexable [run][size].  The
 * group index equals `run` when every run 0..15 occurs (true of the default
 * tables, the only ones the model's encoder uses). */
typedef struct { Code dc[16]; int ndc; Code ac[16][16]; int nac[16]; } EncTables;

static void enc_tables_create(EncTables *t, const HuffSpec *dc, const HuffSpec *ac) {
    Code codes[256];
    memset(t, 0, sizeof *t);
    int n = create_code_table(dc, codes);
    for (int i = 0; i < n; i++) { t->dc[codes[i].data] = codes[i]; if (codes[i].data + 1 > t->ndc) t->ndc = codes[i].data + 1; }
    n = create_code_table(ac, codes);
    for (int i = 0; i < n; i++) {
        int run = (codes[i].data >> 4) & 0xf, size = codes[i].data & 0xf;
        t->ac[run][size] = codes[i];
        if (size + 1 > t->nac[run]) t->nac[run] = size + 1;
    }
}

/* test hooks (G8: jpeg/model/test/test_tables.ml:4-395 prints these tables in full).
 * orc_create_code_table: Specification.create_code_table of any specification, out[3 i + 0 / 1 / 2] = length, bits, data of
 * code i in the order the model's list has them; returns the count.
 * orc_enc_table: Tables.Encoder.dc_table (which 0 luma, 1 chroma) / ac_table (2 luma, 3 chroma) of the default specifications
 * as enc_tables_create builds them.  dc: out[3 i + 0 / 1 / 2] = length, bits, data, returns the entries; ac: out[(16 run +
 * size) * 4 + 0 / 1 / 2 / 3] = length, bits, run, size (a row's missing size-0 symbol is the model's placeholder 0, 0, run 0,
 * size 0), rows[run] = entries of the row, returns 16. */
ORC_API int orc_create_code_table(const int *lengths16, const int *values, int nvalues, int *out) {
    HuffSpec s;
    Code codes[256];
    memset(&s, 0, sizeof s);
    if (nvalues < 0 || nvalues > 256) return -1;
    int total = 0;
    for (int i = 0; i < 16; i++) { s.lengths[i] = lengths16[i]; total += lengths16[i]; }
    if (total > nvalues) return -1;
    for (int i = 0; i < nvalues; i++) s.values[i] = values[i];
    s.nvalues = nvalues;
    int n = create_code_table(&s, codes);
    for (int i = 0; i < n; i++) { out[3 * i] = codes[i].length; out[3 * i + 1] = codes[i].bits; out[3 * i + 2] = codes[i].data; }
    return n;
}
ORC_API int orc_enc_table(int which, int *out, int *rows) {
    EncTables t;
    if (which < 0 || which > 3) return -1;
    enc_tables_create(&t, (which & 1) ? &DC_CHROMA : &DC_LUMA, (which & 1) ? &AC_CHROMA : &AC_LUMA);
    if (which < 2) {
        for (int i = 0; i < t.ndc; i++) { out[3 * i] = t.dc[i].length; out[3 * i + 1] = t.dc[i].bits; out[3 * i + 2] = t.dc[i].data; }
        return t.ndc;
    }
    for (int run = 0; run < 16; run++) {
        rows[run] = t.nac[run];
        for (int size = 0; size < 16; size++) {
            int *o = out + (16 * run + size) * 4;
            o[0] = t.ac[run][size].length; o[1] = t.ac[run][size].bits;
            o[2] = (t.ac[run][size].data >> 4) & 0xf; o[3] = t.ac[run][size].data & 0xf;
        }
    }
    return 16;
}

/* ------------------------------------------------------------------------- */
/* jpeg/model/src/markers.ml                                                  */
typedef struct { int identifier, h, v, tq; } SofComponent;             /* :7-24 */
typedef struct {
    int length, sample_precision, width, height, ncomp;
    SofComponent comp[4];
    int present;
} Sof;                                                                  /* :38-59 */
typedef struct { int selector, dc_sel, ac_sel; } ScanComponent;        /* :75-90 */
typedef struct {
    int length, ncomp;
    ScanComponent comp[4];
    int ss, se, ah, al;
    int present;
} Sos;                                                                  /* :99-129 */
typedef struct { int length, precision, id; int elements[64]; } Dqt;   /* :153-167 */
typedef struct { int length, tclass, id; HuffSpec spec; } Dht;         /* :200-217 */

#define MAX_TABS 16
typedef struct {
    Sof frame;
    Sos scan;
    Dqt dqt[MAX_TABS]; int ndqt;       /* most recent first, like the OCaml list */
    Dht dht[MAX_TABS]; int ndht;
    int restart_interval;              /* -1 = none */
} Header;

#define ORC_E_UNSUPPORTED_MARKER (-3)
#define ORC_E_NO_FRAME_OR_SCAN (-4)
#define ORC_E_NO_COMPONENT (-5)
#define ORC_E_NO_QUANT (-6)
#define ORC_E_NO_HUFF (-7)
#define ORC_E_DC_CODE (-8)
#define ORC_E_AC_CODE (-9)
#define ORC_E_COEF_RANGE (-10)
#define ORC_E_PLANE_OOB (-11)
#define ORC_E_NO_MARKER (-12)
#define ORC_E_TOO_MANY (-13)
#define ORC_E_LUT_OOB (-14) /* Lut.create raises: an over-subscribed Huffman table */

/* decoder.ml:24-29 find_marker.  The OCaml loop never ends on a stream without
 * 0xff (get past the end raises in show only when n >= length); bounded here. */
static int find_marker(Bits *b) {
    bits_align_to_byte(b);
    for (;;) {
        if (b->bit_pos > b->length_in_bits + 64) return ORC_E_NO_MARKER;
        if (bits_get(b, 8) == 0xff) return 0;
    }
}

/* decoder.ml:36-70 Header.decode */
static int header_decode(Bits *b, Header *h) {
    memset(h, 0, sizeof *h);
    h->restart_interval = -1;
    for (;;) {
        int e = find_marker(b);
        if (e) return e;
        int mc = (int)bits_get(b, 8);
        if (mc == 0xc0) { /* sof0: markers.ml:49-59 */
            Sof *s = &h->frame;
            s->length = (int)bits_get(b, 16);
            s->sample_precision = (int)bits_get(b, 8);
            s->height = (int)bits_get(b, 16);
            s->width = (int)bits_get(b, 16);
            s->ncomp = (int)bits_get(b, 8);
            if (s->ncomp > 4) return ORC_E_TOO_MANY;
            for (int i = 0; i < s->ncomp; i++) { /* markers.ml:15-24 */
                s->comp[i].identifier = (int)bits_get(b, 8);
                s->comp[i].h = (int)bits_get(b, 4);
                s->comp[i].v = (int)bits_get(b, 4);
                s->comp[i].tq = (int)bits_get(b, 8);
            }
            s->present = 1;
        } else if (mc == 0xda) { /* sos: markers.ml:111-129 */
            Sos *s = &h->scan;
            s->length = (int)bits_get(b, 16);
            s->ncomp = (int)bits_get(b, 8);
            if (s->ncomp > 4) return ORC_E_TOO_MANY;
            for (int i = 0; i < s->ncomp; i++) {
                s->comp[i].selector = (int)bits_get(b, 8);
                s->comp[i].dc_sel = (int)bits_get(b, 4);
                s->comp[i].ac_sel = (int)bits_get(b, 4);
            }
            s->ss = (int)bits_get(b, 8);
            s->se = (int)bits_get(b, 8);
            s->ah = (int)bits_get(b, 4);
            s->al = (int)bits_get(b, 4);
            s->present = 1;
            return 0;
        } else if (mc == 0xdb) { /* dqt: markers.ml:162-167 (ONE table per segment) */
            if (h->ndqt == MAX_TABS) return ORC_E_TOO_MANY;
            Dqt *q = &h->dqt[h->ndqt++];
            q->length = (int)bits_get(b, 16);
            q->precision = 8 << (int)bits_get(b, 4);
            q->id = (int)bits_get(b, 4);
            for (int i = 0; i < 64; i++) q->elements[i] = (int)bits_get(b, q->precision);
        } else if (mc == 0xc4) { /* dht: markers.ml:209-217 (ONE table per segment) */
            if (h->ndht == MAX_TABS) return ORC_E_TOO_MANY;
            Dht *t = &h->dht[h->ndht++];
            t->length = (int)bits_get(b, 16);
            t->tclass = (int)bits_get(b, 4);
            t->id = (int)bits_get(b, 4);
            int total = 0;
            for (int i = 0; i < 16; i++) { t->spec.lengths[i] = (int)bits_get(b, 8); total += t->spec.lengths[i]; }
            if (total > 256) return ORC_E_TOO_MANY;
            for (int i = 0; i < total; i++) t->spec.values[i] = (int)bits_get(b, 8);
            t->spec.nvalues = total;
        } else if (mc == 0xdd) { /* dri: markers.ml:193-197 */
            (void)bits_get(b, 16);
            h->restart_interval = (int)bits_get(b, 16);
        } else if (mc == 0xd8) { /* soi */
        } else if ((mc >= 0xe0 && mc <= 0xef) || mc == 0xfe) { /* decoder.ml:31-34 skip */
            i64 len = bits_show(b, 16);
            bits_advance(b, len * 8);
        } else {
            return ORC_E_UNSUPPORTED_MARKER;
        }
    }
}

/* ------------------------------------------------------------------------- */
/* decoder.ml:73-87 mag' / mag */
static i64 mag_prime(int cat, i64 code) {
    if (code & ((i64)1 << (cat - 1))) return code;
    return (code | (-((i64)1 << cat))) + 1; /* (code lor (-1 lsl cat)) + 1 */
}
ORC_API i64 orc_mag(int cat, i64 code) { return mag_prime(cat, code); }

typedef struct {
    uint8_t *plane;
    int decoded_width, decoded_height, actual_width, actual_height;
    int x, y;
    i64 dc_pred;
    SofComponent component;
    ScanComponent scan;
    i64 quant_table[64];
    Lut dc_tab, ac_tab;
    i64 coefs[64], dequant[64], idct[64], recon[64];
} Component;                                                           /* decoder.ml:167-187 */

typedef struct orc_decoder {
    Header header;
    Component comp[4];
    int ncomp;
    uint8_t *ecs; i64 ecs_len;
    Bits bits;
    /* iteration state of decode_seq (decoder.ml:374-395) */
    int mb_y, mb_x, id, sy, sx, done;
    i64 blocks_decoded;
} orc_decoder;

/* decoder.ml:261-281 extract_entropy_coded_bits.  Bounded: the OCaml recursion
 * would not terminate on a file with no marker after the scan. */
static int extract_ecs(const Bits *b, uint8_t **out, i64 *out_len) {
    i64 pos = b->bit_pos >> 3;
    i64 cap = b->len - pos + 8; if (cap < 8) cap = 8;
    uint8_t *buf = (uint8_t *)malloc((size_t)cap);
    i64 n = 0;
    int prev = 0;
    for (;;) {
        if (pos > b->len + 4) { free(buf); return ORC_E_NO_MARKER; }
        int c = bits_get_byte(b, pos);
        if (prev == 0xff) {
            if (c == 0x00) { buf[n++] = (uint8_t)prev; pos++; prev = c; }
            else break;
        } else if (c == 0xff) { pos++; prev = c; }
        else { buf[n++] = (uint8_t)c; pos++; prev = c; }
    }
    *out = buf; *out_len = n;
    return 0;
}

static i64 round_up(i64 v, i64 m) { return (v + m - 1) / m * m; }

ORC_API void orc_decoder_destroy(orc_decoder *d) {
    if (!d) return;
    for (int i = 0; i < d->ncomp; i++) { free(d->comp[i].plane); lut_free(&d->comp[i].dc_tab); lut_free(&d->comp[i].ac_tab); }
    free(d->ecs);
    free(d);
}

/* decoder.ml:304-345 init (after Header.decode :36-70) */
ORC_API orc_decoder *orc_decoder_create(const uint8_t *jpg, size_t n, int *err) {
    init_tables();
    orc_decoder *d = (orc_decoder *)calloc(1, sizeof *d);
    Bits b;
    bits_create(&b, jpg, (i64)n);
    int e = header_decode(&b, &d->header);
    if (e) { *err = e; free(d); return NULL; }
    const Header *h = &d->header;
    if (!h->frame.present || !h->scan.present) { *err = ORC_E_NO_FRAME_OR_SCAN; free(d); return NULL; }
    /* :294-302 max_component_scale */
    int max_h = 0, max_v = 0;
    for (int i = 0; i < h->frame.ncomp; i++) {
        if (h->frame.comp[i].h > max_h) max_h = h->frame.comp[i].h;
        if (h->frame.comp[i].v > max_v) max_v = h->frame.comp[i].v;
    }
    /* (a frame whose components all have a zero sampling factor: Int.round_up ~to_multiple_of:0 raises in the model) */
    if (max_h == 0 || max_v == 0) { *err = ORC_E_NO_FRAME_OR_SCAN; free(d); return NULL; }
    i64 rw = round_up(h->frame.width, max_h * 8), rh = round_up(h->frame.height, max_v * 8);
    d->ncomp = h->scan.ncomp;
    for (int i = 0; i < d->ncomp; i++) {
        Component *c = &d->comp[i];
        c->scan = h->scan.comp[i];
        int found = 0; /* :226-230 find_component */
        for (int k = 0; k < h->frame.ncomp; k++)
            if (h->frame.comp[k].identifier == c->scan.selector) { c->component = h->frame.comp[k]; found = 1; break; }
        if (!found) { *err = ORC_E_NO_COMPONENT; d->ncomp = i; orc_decoder_destroy(d); return NULL; }
        c->decoded_width = (int)(rw * c->component.h / max_h);
        c->decoded_height = (int)(rh * c->component.v / max_v);
        c->actual_width = h->frame.width * c->component.h / max_h;
        c->actual_height = h->frame.height * c->component.v / max_v;
        c->plane = (uint8_t *)calloc((size_t)c->decoded_width * c->decoded_height + 1, 1);
        /* :232-236 find_quant_table: the OCaml list is newest-first */
        found = 0;
        for (int k = h->ndqt - 1; k >= 0; k--)
            if (h->dqt[k].id == c->component.tq) { for (int j = 0; j < 64; j++) c->quant_table[j] = h->dqt[k].elements[j]; found = 1; break; }
        if (!found) { *err = ORC_E_NO_QUANT; d->ncomp = i + 1; orc_decoder_destroy(d); return NULL; }
        /* :238-259 huffman tables (newest first) */
        Code codes[256];
        int got = 0;
        for (int k = h->ndht - 1; k >= 0; k--)
            if (h->dht[k].tclass == 0 && h->dht[k].id == c->scan.dc_sel) { int m = create_code_table(&h->dht[k].spec, codes); if (lut_create(&c->dc_tab, codes, m)) got |= 4; got |= 1; break; }
        for (int k = h->ndht - 1; k >= 0; k--)
            if (h->dht[k].tclass == 1 && h->dht[k].id == c->scan.ac_sel) { int m = create_code_table(&h->dht[k].spec, codes); if (lut_create(&c->ac_tab, codes, m)) got |= 4; got |= 2; break; }
        if (got != 3) { *err = (got & 4) ? ORC_E_LUT_OOB : ORC_E_NO_HUFF; d->ncomp = i + 1; orc_decoder_destroy(d); return NULL; }
    }
    e = extract_ecs(&b, &d->ecs, &d->ecs_len);
    if (e) { *err = e; orc_decoder_destroy(d); return NULL; }
    bits_create(&d->bits, d->ecs, d->ecs_len);
    *err = 0;
    return d;
}

/* decoder.ml:89-105 dc_code / ac_code and :118-140 huffman_decode */
static int huffman_decode(Bits *bits, i64 *coefs, const Lut *dc_tab, const Lut *ac_tab) {
    i64 code = bits_show(bits, dc_tab->max_bits);
    if (code < 0) return (int)code;
    if (dc_tab->len[code] == 0) return ORC_E_DC_CODE;
    bits_advance(bits, dc_tab->len[code]);
    int cat = dc_tab->data[code];
    i64 dc = 0;
    /* A DHT may name any byte as a DC category and the model reads that many magnitude bits without a check
     * (decoder.ml:81-96).  Up to 62 everything in mag' is defined (`1 lsl (cat - 1)` and `-1 lsl cat` shift by less than
     * Sys.int_size = 63) and this restatement follows; from 63 on OCaml leaves the shifts unspecified -- there is no
     * model result, the product refuses the stream (include/hvc_jpeg.h), and so does this checker.  (A category of as
     * many bits as the segment has, or more, raises in Bits.get either way.) */
    if (cat > 62) return cat >= bits->length_in_bits ? ORC_E_BITS_OOB : ORC_E_DC_CODE;
    if (cat != 0) { i64 v = bits_get(bits, cat); if (v < 0) return (int)v; dc = mag_prime(cat, v); }
    coefs[0] = dc;
    int cof_cnt = 1;
    while (cof_cnt < 64) {
        code = bits_show(bits, ac_tab->max_bits);
        if (code < 0) return (int)code;
        if (ac_tab->len[code] == 0) return ORC_E_AC_CODE;
        bits_advance(bits, ac_tab->len[code]);
        int run = (ac_tab->data[code] >> 4) & 0xf, size = ac_tab->data[code] & 0xf;
        i64 mag = 0;
        if (size != 0) { i64 v = bits_get(bits, size); if (v < 0) return (int)v; mag = mag_prime(size, v); }
        if (mag == 0 && run == 0) cof_cnt = 64;
        else {
            cof_cnt += run;
            if (cof_cnt >= 64) return ORC_E_COEF_RANGE;
            coefs[cof_cnt] = mag;
            cof_cnt++;
        }
    }
    return 0;
}

/* decoder.ml:347-360 decode_block (with :151-165 decode_coefficient_block) */
static int decode_block(orc_decoder *d, Component *c) {
    for (int i = 0; i < 64; i++) c->coefs[i] = 0; /* clear_block :112-116 */
    int e = huffman_decode(&d->bits, c->coefs, &c->dc_tab, &c->ac_tab);
    if (e) return e;
    c->dc_pred = orc_dequantize_dc_pred_and_inverse_zigzag(c->quant_table, c->dc_pred, c->coefs, c->dequant);
    memcpy(c->idct, c->dequant, sizeof c->idct);
    orc_idct_8x8(c->idct);
    if (recon_block(c->idct, c->recon, c->plane, c->decoded_width, c->decoded_height,
                    (size_t)c->decoded_width, c->x, c->y))
        return ORC_E_PLANE_OOB;
    return 0;
}

/* One step of For_testing.Sequenced.decode (decoder.ml:433-435) = one element
 * of decode_seq (:374-395, with decode_component_seq :362-372): loops mcu_y,
 * mcu_x, component id, y<vscale, x<hscale.  Returns component index >= 0,
 * -1 when the sequence is exhausted, < -1 on error. */
ORC_API int orc_decoder_next_block(orc_decoder *d) {
    if (d->done) return -1;
    const Component *c0 = &d->comp[0];
    if (c0->component.h == 0 || c0->component.v == 0) return ORC_E_PLANE_OOB; /* (Division_by_zero in decode_seq :374-380) */
    int mbs_wide = c0->decoded_width / (8 * c0->component.h);
    int mbs_high = c0->decoded_height / (8 * c0->component.v);
    if (mbs_wide == 0 || mbs_high == 0) { d->done = 1; return -1; }
    Component *c = &d->comp[d->id];
    int hscale = c->component.h, vscale = c->component.v;
    c->x = ((d->mb_x * hscale) + d->sx) * 8;
    c->y = ((d->mb_y * vscale) + d->sy) * 8;
    int e = decode_block(d, c);
    if (e) return e;
    d->blocks_decoded++;
    int id = d->id;
    /* advance the nested iteration */
    if (++d->sx >= hscale) { d->sx = 0;
        if (++d->sy >= vscale) { d->sy = 0;
            if (++d->id >= d->ncomp) { d->id = 0;
                if (++d->mb_x >= mbs_wide) { d->mb_x = 0;
                    if (++d->mb_y >= mbs_high) d->done = 1; } } } }
    /* zero-sized components would loop forever in this form; the model's
     * Sequence.init 0 just yields nothing.  Skip them. */
    while (!d->done && (d->comp[d->id].component.h == 0 || d->comp[d->id].component.v == 0)) {
        if (++d->id >= d->ncomp) { d->id = 0; if (++d->mb_x >= mbs_wide) { d->mb_x = 0; if (++d->mb_y >= mbs_high) d->done = 1; } }
    }
    return id;
}

/* decoder.ml:397 decode */
ORC_API int orc_decoder_decode(orc_decoder *d) {
    for (;;) {
        int r = orc_decoder_next_block(d);
        if (r == -1) return 0;
        if (r < -1) return r;
    }
}

/* Test harness around For_testing.Sequenced.decode (decoder.ml:433-435), no arithmetic of its own: runs the
 * sequence to its end and collects every block's `coefs` (zig-zag order, decoder.ml:118-140) with coefs[0]
 * replaced by the component's dc_pred after the block -- the absolute DC of decoder.ml:143 -- in the C ABI's
 * record layout: component planes back to back, each [decoded_height/8][decoded_width/8][64].  int64, so a DC
 * outside int16 (the model is 63-bit) is representable.  The planes are decoded as a side effect.  Returns 0,
 * or the model's error (< -1) with the blocks decoded so far in place. */
ORC_API int orc_decoder_coef_record(orc_decoder *d, i64 *out) {
    size_t base[4], at = 0;
    for (int i = 0; i < d->ncomp; i++) {
        base[i] = at;
        at += (size_t)(d->comp[i].decoded_width / 8) * (size_t)(d->comp[i].decoded_height / 8) * 64;
    }
    for (;;) {
        int r = orc_decoder_next_block(d);
        if (r == -1) return 0;
        if (r < -1) return r;
        const Component *c = &d->comp[r];
        i64 *blk = out + base[r] + ((size_t)(c->y / 8) * (size_t)(c->decoded_width / 8) + (size_t)(c->x / 8)) * 64;
        memcpy(blk, c->coefs, sizeof c->coefs);
        blk[0] = c->dc_pred;
    }
}
ORC_API i64 orc_decoder_coef_count(const orc_decoder *d) {
    i64 n = 0;
    for (int i = 0; i < d->ncomp; i++) n += (i64)(d->comp[i].decoded_width / 8) * (d->comp[i].decoded_height / 8) * 64;
    return n;
}

ORC_API int orc_decoder_ncomp(const orc_decoder *d) { return d->ncomp; }
ORC_API int orc_decoder_width(const orc_decoder *d) { return d->header.frame.width; }
ORC_API int orc_decoder_height(const orc_decoder *d) { return d->header.frame.height; }

/* info[12]: decoded_w, decoded_h, actual_w, actual_h, x, y, dc_pred, identifier,
 * hscale, vscale, tq, 0 */
ORC_API void orc_decoder_component_info(const orc_decoder *d, int i, i64 *info) {
    const Component *c = &d->comp[i];
    info[0] = c->decoded_width; info[1] = c->decoded_height;
    info[2] = c->actual_width; info[3] = c->actual_height;
    info[4] = c->x; info[5] = c->y; info[6] = c->dc_pred; info[7] = c->component.identifier;
    info[8] = c->component.h; info[9] = c->component.v; info[10] = c->component.tq; info[11] = 0;
}
/* which: 0 coefs (zig-zag, DC differential), 1 dequant, 2 idct (clipped), 3 recon, 4 quant_table */
ORC_API const i64 *orc_decoder_component_array(const orc_decoder *d, int i, int which) {
    const Component *c = &d->comp[i];
    switch (which) {
    case 0: return c->coefs; case 1: return c->dequant; case 2: return c->idct;
    case 3: return c->recon; default: return c->quant_table;
    }
}
/* decoder.ml:399-401 get_decoded_planes */
ORC_API const uint8_t *orc_decoder_plane(const orc_decoder *d, int i) { return d->comp[i].plane; }

/* decoder.ml:403-413 crop (Plane.blit_available top-left, plane.ml:22-35) */
ORC_API void orc_decoder_cropped_plane(const orc_decoder *d, int i, uint8_t *out) {
    const Component *c = &d->comp[i];
    for (int r = 0; r < c->actual_height; r++)
        memcpy(out + (size_t)r * c->actual_width, c->plane + (size_t)r * c->decoded_width, (size_t)c->actual_width);
}

/* decoder.ml:415-420 get_yuv_frame = Frame.of_planes ~y:(crop components.(0)) ~u:(crop components.(1)) ~v:(crop components.(2))
 * with common/src/frame.ml:42-61 of_planes / infer_chroma_subsampling and :3-22 Chroma_subsampling.width / height.
 * Returns 420 / 422 / 444, or < 0 where the model raises: components.(1) or .(2) missing (Invalid_argument "index out of
 * bounds"), "Chroma planes must be same width and height", "Could not infer chroma subsampling". */
#define ORC_E_FRAME_INDEX (-15)
#define ORC_E_FRAME_CHROMA_SIZES (-16)
#define ORC_E_FRAME_INFER (-17)
ORC_API int orc_decoder_frame_of_planes(const orc_decoder *d) {
    if (d->ncomp < 3) return ORC_E_FRAME_INDEX;
    const Component *y = &d->comp[0], *u = &d->comp[1], *v = &d->comp[2];
    if (u->actual_width != v->actual_width || u->actual_height != v->actual_height) return ORC_E_FRAME_CHROMA_SIZES;
    /* check C420, then C422, then C444 (frame.ml:50-59) */
    if (y->actual_width / 2 == u->actual_width && y->actual_height / 2 == u->actual_height) return 420;
    if (y->actual_width / 2 == u->actual_width && y->actual_height == u->actual_height) return 422;
    if (y->actual_width == u->actual_width && y->actual_height == u->actual_height) return 444;
    return ORC_E_FRAME_INFER;
}

/* ------------------------------------------------------------------------- */
/* jpeg/model/src/encoder.ml                                                  */

/* :143 size ; :145-147 magnitude */
static int enc_size(i64 v) {
    if (v == 0) return 0;
    i64 a = v < 0 ? -v : v;
    int n = 0;
    while (a) { n++; a >>= 1; }
    return n; /* floor_log2 |v| + 1 */
}
static i64 enc_magnitude(int size, i64 v) {
    i64 mask = ((i64)1 << size) - 1;
    return v >= 0 ? (v & mask) : ((v - 1) & mask);
}
ORC_API int orc_enc_size(i64 v) { return enc_size(v); }
ORC_API i64 orc_enc_magnitude(int size, i64 v) { return enc_magnitude(size, v); }

typedef struct { int run; i64 value; } Rle;

/* :127-141 rle.  Returns count; out[0] is the DC difference entry. */
static int enc_rle(const i64 *quant, i64 *dc_pred, Rle *out) {
    int n = 0;
    i64 dc = quant[0];
    out[n].run = 0; out[n].value = dc - *dc_pred; n++;
    int run = 0;
    for (int pos = 1; pos <= 63; pos++) {
        i64 value = quant[pos];
        if (pos == 63) { out[n].run = run; out[n].value = value; n++; }
        else if (value != 0) { out[n].run = run; out[n].value = value; n++; run = 0; }
        else run++;
    }
    *dc_pred = dc;
    return n;
}
/* test hook: rle of one block -> (run,value) pairs */
ORC_API int orc_enc_rle(const i64 *quant, i64 dc_pred, int *runs, i64 *values) {
    Rle r[65];
    int n = enc_rle(quant, &dc_pred, r);
    for (int i = 0; i < n; i++) { runs[i] = r[i].run; values[i] = r[i].value; }
    return n;
}

/* :149-193 write_bits */
static void enc_write_bits(Writer *w, const Rle *rle, int n, const EncTables *t) {
    /* write_dc */
    {
        i64 value = rle[0].value;
        int size = enc_size(value);
        Code code = t->dc[size];
        w_put_bits(w, 1, code.bits, code.length);
        w_put_bits(w, 1, enc_magnitude(size, value), size);
    }
    for (int i = 1; i < n; i++) {
        int run = rle[i].run;
        i64 value = rle[i].value;
        if (i == n - 1 && value == 0) { /* [ {run; value = 0} ] -> end of block */
            Code code = t->ac[0][0];
            w_put_bits(w, 1, code.bits, code.length);
            break;
        }
        while (run >= 16) { /* runs: write_ac 15 0 */
            Code code = t->ac[15][0];
            w_put_bits(w, 1, code.bits, code.length);
            run -= 16;
        }
        int size = enc_size(value);
        Code code = t->ac[run][size];
        w_put_bits(w, 1, code.bits, code.length);
        w_put_bits(w, 1, enc_magnitude(size, value), size);
    }
}

static void write_marker_code(Writer *w, int code) { /* :207-210 */
    w_put_bits(w, 0, 0xff, 8);
    w_put_bits(w, 0, code, 8);
}
/* :212-222 + markers.ml:219-231 */
static void write_dht(Writer *w, int tclass, int id, const HuffSpec *s) {
    write_marker_code(w, 0xc4);
    int total = 0;
    for (int i = 0; i < 16; i++) total += s->lengths[i];
    w_put_bits(w, 0, 3 + 16 + total, 16);
    w_put_bits(w, 0, tclass, 4);
    w_put_bits(w, 0, id, 4);
    for (int i = 0; i < 16; i++) w_put_bits(w, 0, s->lengths[i], 8);
    for (int i = 0; i < total; i++) w_put_bits(w, 0, s->values[i], 8);
}
/* :224-229 + markers.ml:169-183 (element_precision = 8) */
static void write_dqt(Writer *w, int id, const int *q) {
    write_marker_code(w, 0xdb);
    w_put_bits(w, 0, 3 + 64, 16);
    w_put_bits(w, 0, 0, 4);
    w_put_bits(w, 0, id, 4);
    for (int i = 0; i < 64; i++) w_put_bits(w, 0, q[i], 8);
}

typedef struct { int quant_table, dc_tab, ac_tab, component, h, v; } ScanParam; /* :288-295 */

/* :371-418 write_headers for Parameters.yuv (:306-345) / monochrome (:351-368) */
static void write_headers(Writer *w, int width, int height, const int *qluma, const int *qchroma,
                          const ScanParam *sc, int nsc) {
    write_marker_code(w, 0xd8);
    { /* :231-237 write_app0 "Hardcaml JPEG." */
        const char *data = "Hardcaml JPEG.";
        write_marker_code(w, 0xe0);
        w_put_bits(w, 0, 2 + (i64)strlen(data), 16);
        for (size_t i = 0; i < strlen(data); i++) w_put_bits(w, 0, (unsigned char)data[i], 8);
    }
    write_dqt(w, 0, qluma);
    if (nsc > 1) write_dqt(w, 1, qchroma);
    /* :239-250 write_sof + markers.ml:61-71 */
    write_marker_code(w, 0xc0);
    w_put_bits(w, 0, 2 + 6 + nsc * 3, 16);
    w_put_bits(w, 0, 8, 8);
    w_put_bits(w, 0, height, 16);
    w_put_bits(w, 0, width, 16);
    w_put_bits(w, 0, nsc, 8);
    for (int i = 0; i < nsc; i++) {
        w_put_bits(w, 0, sc[i].component, 8);
        w_put_bits(w, 0, sc[i].h, 4);
        w_put_bits(w, 0, sc[i].v, 4);
        w_put_bits(w, 0, sc[i].quant_table, 8);
    }
    write_dht(w, 0, 0, &DC_LUMA);
    if (nsc > 1) write_dht(w, 0, 1, &DC_CHROMA);
    write_dht(w, 1, 0, &AC_LUMA);
    if (nsc > 1) write_dht(w, 1, 1, &AC_CHROMA);
    /* :252-264 write_sos + markers.ml:131-150 */
    write_marker_code(w, 0xda);
    w_put_bits(w, 0, 2 + 4 + nsc * 2, 16);
    w_put_bits(w, 0, nsc, 8);
    for (int i = 0; i < nsc; i++) {
        w_put_bits(w, 0, sc[i].component, 8);
        w_put_bits(w, 0, sc[i].dc_tab, 4);
        w_put_bits(w, 0, sc[i].ac_tab, 4);
    }
    w_put_bits(w, 0, 0, 8);
    w_put_bits(w, 0, 63, 8);
    w_put_bits(w, 0, 0, 4);
    w_put_bits(w, 0, 0, 4);
}

static void scan_params(int chroma, ScanParam *sc, int *nsc) {
    /* Parameters.c420/c422/c444 (:347-349): scales [|hY;vY;hU;vU;hV;vV|] */
    static const int S420[6] = {2, 2, 1, 1, 1, 1}, S422[6] = {2, 2, 1, 2, 1, 2}, S444[6] = {1, 1, 1, 1, 1, 1};
    const int *s = chroma == 420 ? S420 : chroma == 422 ? S422 : S444;
    if (chroma == 400) { /* monochrome :351-368 */
        sc[0] = (ScanParam){0, 0, 0, 1, 1, 1};
        *nsc = 1;
        return;
    }
    sc[0] = (ScanParam){0, 0, 0, 1, s[0], s[1]};
    sc[1] = (ScanParam){1, 1, 1, 2, s[2], s[3]};
    sc[2] = (ScanParam){1, 1, 1, 3, s[4], s[5]};
    *nsc = 3;
}

/* Header only: Encoder.write_headers ~params:(Parameters.cXXX ~width ~height ~quality) */
ORC_API i64 orc_write_headers(int width, int height, int chroma, int quality, uint8_t *out, size_t cap) {
    Writer w; w_init(&w);
    int ql[64], qc[64];
    ScanParam sc[3]; int nsc;
    orc_quant_scale(Q_LUMA, quality, ql);
    orc_quant_scale(Q_CHROMA, quality, qc);
    scan_params(chroma, sc, &nsc);
    write_headers(&w, width, height, ql, qc, sc, nsc);
    i64 n = (i64)w.n;
    if (w.n <= cap) memcpy(out, w.buffer, w.n); else n = -1;
    free(w.buffer);
    return n;
}

/* encoder.ml:512-541 encode_yuv / encode_420/422/444 (and :543-551 monochrome).
 * y/u/v: tight planes of the frame (Frame.create sizes, frame.ml:33-41).
 * If coef_out != NULL it receives, per scan component in order, the quantised
 * blocks in block-raster order [by][bx][64] (zig-zag, DC absolute): the layout
 * of the C-ABI, for GPU parity tests.  Returns bytes written or <0. */
ORC_API i64 orc_encode_yuv(const uint8_t *y, const uint8_t *u, const uint8_t *v, int width,
                           int height, int chroma, int quality, uint8_t *out, size_t cap,
                           int16_t *coef_out) {
    init_tables();
    Writer w; w_init(&w);
    int ql[64], qc[64];
    ScanParam sc[3]; int nsc;
    orc_quant_scale(Q_LUMA, quality, ql);
    orc_quant_scale(Q_CHROMA, quality, qc);
    scan_params(chroma, sc, &nsc);
    EncTables et[2];
    enc_tables_create(&et[0], &DC_LUMA, &AC_LUMA);
    enc_tables_create(&et[1], &DC_CHROMA, &AC_CHROMA);
    /* create (:437-472) */
    int max_h = 0, max_v = 0;
    for (int i = 0; i < nsc; i++) { if (sc[i].h > max_h) max_h = sc[i].h; if (sc[i].v > max_v) max_v = sc[i].v; }
    uint8_t *planes[3] = {0, 0, 0};
    int pw[3] = {0, 0, 0}, ph[3] = {0, 0, 0};
    const uint8_t *src[3] = {y, u, v};
    int cw = chroma == 444 || chroma == 400 ? width : width / 2;     /* frame.ml:10-24 */
    int chh = chroma == 420 ? height / 2 : height;
    int sw[3] = {width, cw, cw}, sh[3] = {height, chh, chh};
    for (int i = 0; i < nsc; i++) {
        i64 wd = (i64)width * sc[i].h / max_h, ht = (i64)height * sc[i].v / max_v;
        pw[i] = (int)round_up(wd, 8 * sc[i].h);
        ph[i] = (int)round_up(ht, 8 * sc[i].v);
        planes[i] = (uint8_t *)calloc((size_t)pw[i] * ph[i] + 1, 1); /* zero-filled: plane.ml:11-17 */
        /* Plane.blit_available (:514-516; plane.ml:22-35) */
        int bwid = sw[i] < pw[i] ? sw[i] : pw[i], bh = sh[i] < ph[i] ? sh[i] : ph[i];
        for (int r = 0; r < bh; r++) memcpy(planes[i] + (size_t)r * pw[i], src[i] + (size_t)r * sw[i], (size_t)bwid);
    }
    write_headers(&w, width, height, ql, qc, sc, nsc);
    /* encode_seq (:476-505) */
    int mbs_wide = pw[0] / (8 * sc[0].h), mbs_high = ph[0] / (8 * sc[0].v);
    i64 dc_pred[3] = {0, 0, 0};
    size_t coef_base[3]; size_t acc = 0;
    for (int i = 0; i < nsc; i++) { coef_base[i] = acc; acc += (size_t)(pw[i] / 8) * (ph[i] / 8) * 64; }
    i64 fd[64], qu[64], qt[64];
    Rle rle[65];
    for (int y_mb = 0; y_mb < mbs_high; y_mb++)
        for (int x_mb = 0; x_mb < mbs_wide; x_mb++)
            for (int i = 0; i < nsc; i++)
                for (int ys = 0; ys < sc[i].v; ys++)
                    for (int xs = 0; xs < sc[i].h; xs++) {
                        int x_blk = x_mb * sc[i].h + xs, y_blk = y_mb * sc[i].v + ys;
                        if (x_blk * 8 + 8 > pw[i] || y_blk * 8 + 8 > ph[i]) {
                            /* "[Plane.get] out of bounds" (plane.ml:43-50 via encoder.ml:85): the MCU grid of
                             * component 0 reaches past this component's plane (e.g. 4:2:0 at width 16k + 1) */
                            free(w.buffer);
                            for (int k = 0; k < nsc; k++) free(planes[k]);
                            return -13;
                        }
                        const int *tab = sc[i].quant_table == 0 ? ql : qc;
                        for (int k = 0; k < 64; k++) qt[k] = tab[k];
                        /* encode_block :195-205 */
                        encode_block_stage(planes[i], (size_t)pw[i], x_blk * 8, y_blk * 8, qt, fd, qu);
                        if (coef_out) {
                            int16_t *dst = coef_out + coef_base[i] + ((size_t)y_blk * (pw[i] / 8) + x_blk) * 64;
                            for (int k = 0; k < 64; k++) dst[k] = (int16_t)qu[k];
                        }
                        int n = enc_rle(qu, &dc_pred[i], rle);
                        enc_write_bits(&w, rle, n, &et[sc[i].dc_tab]);
                    }
    /* complete_and_write_eoi :507-510 */
    w_flush_with_1s(&w, 1);
    write_marker_code(&w, 0xd9);
    i64 n = (i64)w.n;
    if (w.n <= cap) memcpy(out, w.buffer, w.n); else n = -1;
    free(w.buffer);
    for (int i = 0; i < nsc; i++) free(planes[i]);
    return n;
}

/* ------------------------------------------------------------------------- */
/* tools/src/planar_444.ml                                                    */
static inline int avg2(int a, int b) { return (a + b + 1) >> 1; }                 /* :4-8 */
static inline int avg4(int a, int b, int c, int d) { return (a + b + c + d + 2) >> 2; } /* :10-16 */

/* :82-103 supersample_hv2 over all rows (:122-131 convert_from_420): src w x h
 * -> dst 2w x 2h */
ORC_API void orc_supersample_hv2(const uint8_t *src, int w, int h, uint8_t *dst) {
    int dw = 2 * w;
    for (int row = 0; row < h; row++) {
        int row1 = row, row2 = row + 1 < h - 1 ? row + 1 : h - 1;
        for (int col = 0; col <= w - 2; col++) {
            int a = src[row1 * w + col], b = src[row1 * w + col + 1];
            int c = src[row2 * w + col], d = src[row2 * w + col + 1];
            dst[(row * 2) * dw + col * 2] = (uint8_t)a;
            dst[(row * 2) * dw + col * 2 + 1] = (uint8_t)avg2(a, b);
            dst[(row * 2 + 1) * dw + col * 2] = (uint8_t)avg2(a, c);
            dst[(row * 2 + 1) * dw + col * 2 + 1] = (uint8_t)avg4(a, b, c, d);
        }
        int a = src[row1 * w + w - 1], b = src[row2 * w + w - 1];
        dst[(row * 2) * dw + w * 2 - 2] = (uint8_t)a;
        dst[(row * 2) * dw + w * 2 - 1] = (uint8_t)a;
        dst[(row * 2 + 1) * dw + w * 2 - 2] = (uint8_t)avg2(a, b);
        dst[(row * 2 + 1) * dw + w * 2 - 1] = (uint8_t)avg2(a, b);
    }
}
/* :69-80 subsample_hv2 (:105-114 convert_to_420): src 2w x 2h(+) -> dst w x h ;
 * sw = source width */
ORC_API void orc_subsample_hv2(const uint8_t *src, int sw, uint8_t *dst, int w, int h) {
    for (int row = 0; row < h; row++)
        for (int col = 0; col < w; col++)
            dst[row * w + col] = (uint8_t)avg4(src[(row * 2) * sw + col * 2], src[(row * 2) * sw + col * 2 + 1],
                                               src[(row * 2 + 1) * sw + col * 2], src[(row * 2 + 1) * sw + col * 2 + 1]);
}
/* :25-33 supersample_h2 (:52-61 convert_from_422) */
ORC_API void orc_supersample_h2(const uint8_t *src, int w, int h, uint8_t *dst) {
    int dw = 2 * w;
    for (int row = 0; row < h; row++) {
        for (int col = 0; col <= w - 2; col++) {
            dst[row * dw + col * 2] = src[row * w + col];
            dst[row * dw + col * 2 + 1] = (uint8_t)avg2(src[row * w + col], src[row * w + col + 1]);
        }
        dst[row * dw + w * 2 - 2] = src[row * w + w - 1];
        dst[row * dw + w * 2 - 1] = src[row * w + w - 1];
    }
}
/* :18-23 subsample_h2 (:35-44 convert_to_422) */
ORC_API void orc_subsample_h2(const uint8_t *src, int sw, uint8_t *dst, int w, int h) {
    for (int row = 0; row < h; row++)
        for (int col = 0; col < w; col++)
            dst[row * w + col] = (uint8_t)avg2(src[row * sw + col * 2], src[row * sw + col * 2 + 1]);
}
/* tools/src/yuv.ml:43-61 crop (one plane; clamped source coordinates) */
ORC_API void orc_crop_plane(const uint8_t *src, int sw, int sh, int x_pos, int y_pos, uint8_t *dst, int dw, int dh) {
    for (int r = 0; r < dh; r++)
        for (int c = 0; c < dw; c++) {
            int col = c + x_pos; col = col < 0 ? 0 : (col >= sw ? sw - 1 : col);
            int row = r + y_pos; row = row < 0 ? 0 : (row >= sh ? sh - 1 : row);
            dst[r * dw + c] = src[row * sw + col];
        }
}

/* tools/src/packed_422.ml:6-8 : byte offsets of Y, U, V inside a group of four bytes (two pixels); the second luma
 * sample sits at yo + 2.  which: 1 = yuy2, 2 = uyvy, 3 = yvyu (the numbers of include/hvc_jpeg.h) */
static void packed_fmt(int which, int *yo, int *uo, int *vo) {
    if (which == 1) { *yo = 0; *uo = 1; *vo = 3; }
    else if (which == 2) { *yo = 1; *uo = 0; *vo = 2; }
    else { *yo = 0; *uo = 3; *vo = 1; }
}
/* :10-23 convert_to_planar : src is a plane of (2 w) x h bytes, dst a 4:2:2 frame of luma size w x h */
ORC_API void orc_packed422_to_planar(int which, const uint8_t *src, int w, int h, uint8_t *y, uint8_t *u, uint8_t *v) {
    int yo, uo, vo;
    packed_fmt(which, &yo, &uo, &vo);
    for (int row = 0; row < h; row++)
        for (int col = 0; col < w / 2; col++) {
            const uint8_t *p = src + (size_t)row * (2 * w) + col * 4;
            y[(size_t)row * w + col * 2] = p[yo];
            y[(size_t)row * w + col * 2 + 1] = p[yo + 2];
            u[(size_t)row * (w / 2) + col] = p[uo];
            v[(size_t)row * (w / 2) + col] = p[vo];
        }
}
/* :33-46 convert_from_planar */
ORC_API void orc_packed422_from_planar(int which, const uint8_t *y, const uint8_t *u, const uint8_t *v, int w, int h, uint8_t *dst) {
    int yo, uo, vo;
    packed_fmt(which, &yo, &uo, &vo);
    for (int row = 0; row < h; row++)
        for (int col = 0; col < w / 2; col++) {
            uint8_t *p = dst + (size_t)row * (2 * w) + col * 4;
            p[yo] = y[(size_t)row * w + col * 2];
            p[yo + 2] = y[(size_t)row * w + col * 2 + 1];
            p[uo] = u[(size_t)row * (w / 2) + col];
            p[vo] = v[(size_t)row * (w / 2) + col];
        }
}

/* tools/src/ocompare.ml:8-19 max_difference ; :41-52 square_error */
ORC_API i64 orc_max_difference(const uint8_t *a, const uint8_t *b, i64 n) {
    i64 m = 0;
    for (i64 i = 0; i < n; i++) { i64 d = a[i] > b[i] ? a[i] - b[i] : b[i] - a[i]; if (d > m) m = d; }
    return m;
}
ORC_API i64 orc_square_error(const uint8_t *a, const uint8_t *b, i64 n) {
    i64 acc = 0;
    for (i64 i = 0; i < n; i++) { i64 d = (i64)a[i] - (i64)b[i]; acc += d * d; }
    return acc;
}
