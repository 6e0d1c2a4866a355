// hvc_kernels.hip -- gfx950 (MI355X, CDNA4) kernels of the JPEG block-transform path.
//
// Decode (K1): dequantise + inverse zig-zag + Chen-Wang integer IDCT + clip/level
// shift + plane store, bit-exact to hardcamls/video-coding's OCaml model
//   jpeg/model/src/decoder.ml:142-149, 213-224, 347-360; jpeg/model/src/dct.ml:11-107.
//
// Mapping: ONE 8x8 BLOCK PER LANE, all 64 values in VGPRs.
//  * the inverse zig-zag is a compile-time register renaming (no LDS, no scatter);
//  * the row pass and the column pass need no transpose at all (a lane owns the
//    whole block), so there is no cross-lane traffic and every VALU lane does
//    useful butterfly work (the 16-lanes-per-block even/odd split would idle half
//    of every VALU instruction);
//  * a wave covers 64 consecutive blocks: its loads are one contiguous 8 KiB run
//    of the coefficient plane and each of its 8 row stores is 64 lanes x 8 B =
//    512 contiguous bytes of a pixel row.
//
// Arithmetic: the model computes in OCaml's 63-bit ints.  The fast kernel
// computes in int32 with 24-bit multiplies (v_mul_i32_i24 / v_mad_i32_i24, full
// rate) under an exact in-situ range guard; a block whose intermediates could
// leave the proven-safe ranges is appended to a fix-up list and recomputed by
// the wide kernel in int64, so results are exact for every int16 input.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hvc_kernels.h"

#ifndef HVC_FAST_LB
#define HVC_FAST_LB HVC_TILE /* experiments: -DHVC_FAST_LB="HVC_TILE,6" forces 6 waves per SIMD */
#endif

namespace hvc {

// jpeg/model/src/zigzag.ml:3-69  inverse[zz] = raster
__device__ constexpr int ZI[64] = {
    0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,
    12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
    58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
// jpeg/model/src/zigzag.ml:71-137  forward[raster] = zz
__device__ constexpr int ZF[64] = {
    0,  1,  5,  6,  14, 15, 27, 28, 2,  4,  7,  13, 16, 26, 29, 42,
    3,  8,  12, 17, 25, 30, 41, 43, 9,  11, 18, 24, 31, 40, 44, 53,
    10, 19, 23, 32, 39, 45, 52, 54, 20, 22, 33, 38, 46, 51, 55, 60,
    21, 34, 37, 47, 50, 56, 59, 61, 35, 36, 48, 49, 57, 58, 62, 63};

// dct.ml:4-9
constexpr int W1 = 2841, W2 = 2676, W3 = 2408, W5 = 1609, W6 = 1108, W7 = 565;

// ---------------------------------------------------------------------------
// Range guard of the int32 fast path (proved by tests/test_guard_bounds.py with
// interval arithmetic over exactly the operations below):
//   |dequantised coefficient|        <= GUARD_D   (via the coefficient energy, below)
//   |row-pass output|                <= GUARD_R
//   |argument of a 181*y product|    <= GUARD_Y   (row and column pass)
// imply that no int32 operation wraps and every v_mul_i32_i24 / v_mad_i32_i24
// operand lies in [-2^23, 2^23).
constexpr int GUARD_D = HVC_GUARD_D;
constexpr int GUARD_R = (1 << 18) - 1;
constexpr int GUARD_Y = (1 << 23) - 1;

struct Guard {
    // |dequantised coefficient| is bounded a priori: every |c[k] * q[k]| <= qmax * sqrt(E) with
    // E = sum c[k]^2 accumulated by v_dot2_i32_i16 (saturating) straight from the packed
    // coefficient dwords; the host passes ethr = (GUARD_D / qmax)^2 per table.
    int energy = 0;
    int rmax = 0, rmin = 0, ymax = 0, ymin = 0;
    __device__ __forceinline__ void e2(unsigned packed_pair) {
        typedef short short2v __attribute__((ext_vector_type(2)));
        short2v pr = __builtin_bit_cast(short2v, packed_pair);
        energy = __builtin_amdgcn_sdot2(pr, pr, energy, true);
    }
    __device__ __forceinline__ void r2(int a, int b) { rmax = max(max(rmax, a), b); rmin = min(min(rmin, a), b); }
    __device__ __forceinline__ void y2(int a, int b) { ymax = max(max(ymax, a), b); ymin = min(min(ymin, a), b); }
    __device__ __forceinline__ bool bad(int ethr) const {
        return (energy > ethr) | (rmax > GUARD_R) | (rmin < -GUARD_R) | (ymax > GUARD_Y) | (ymin < -GUARD_Y);
    }
};

// dequantise one half of a packed coefficient pair: the sign extension of the int16 half rides in
// the SDWA source selector of the 24-bit multiply (one instruction, no unpack).  q is wave-uniform.
__device__ __forceinline__ int dequant_lo(unsigned pair, int q) {
    int d;
    asm("v_mul_i32_i24_sdwa %0, sext(%1), %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD"
        : "=v"(d) : "v"(pair), "s"(q));
    return d;
}
__device__ __forceinline__ int dequant_hi(unsigned pair, int q) {
    int d;
    asm("v_mul_i32_i24_sdwa %0, sext(%1), %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD"
        : "=v"(d) : "v"(pair), "s"(q));
    return d;
}

__device__ __forceinline__ int mul24(int a, int b) { return __mul24(a, b); }
__device__ __forceinline__ int mad24(int a, int b, int c) { return __mul24(a, b) + c; }

// One 1-D Chen-Wang pass in int32 (dct.ml:11-54 row / :56-98 column).
//   COL = false: x0 = b0<<11 + 128, no rotation rounding, output >> 8
//   COL = true : x0 = b0<<8 + 8192 (+ BIAS), rotations (+4)>>3, output >> 14
// The rotations are written in the expanded form  W1*x4 + W7*x5  ==
// W7*(x4+x5) + (W1-W7)*x4  (identical integers; operands stay within 24 bits).
// (the column pass leaves its outputs UNSHIFTED: the >> 14 is done by the
// saturating pack instruction of the store stage.)
template <bool COL, int BIAS>
__device__ __forceinline__ void idct_1d_fast(int &b0, int &b1, int &b2, int &b3, int &b4, int &b5,
                                             int &b6, int &b7, Guard &g) {
    constexpr int R = COL ? 4 : 0;
    int x0 = COL ? (b0 << 8) + (8192 + BIAS) : (b0 << 11) + 128;
    int x1 = COL ? (b4 << 8) : (b4 << 11);
    int x2 = b6, x3 = b2, x4 = b1, x5 = b7, x6 = b5, x7 = b3;
    // first stage
    int n4 = mad24(W7, x5, mad24(W1, x4, R));
    int n5 = mad24(-W1, x5, mad24(W7, x4, R));
    int n6 = mad24(W3, x7, mad24(W5, x6, R));
    int n7 = mad24(-W5, x7, mad24(W3, x6, R));
    // second stage
    int n3 = mad24(W6, x2, mad24(W2, x3, R));
    int n2 = mad24(-W2, x2, mad24(W6, x3, R));
    if (COL) { n4 >>= 3; n5 >>= 3; n6 >>= 3; n7 >>= 3; n3 >>= 3; n2 >>= 3; }
    int x8 = x0 + x1;
    x0 = x0 - x1;
    x1 = n4 + n6;
    x4 = n4 - n6;
    x6 = n5 + n7;
    x5 = n5 - n7;
    // third stage
    x7 = x8 + n3;
    x8 = x8 - n3;
    x3 = x0 + n2;
    x0 = x0 - n2;
    int ys = x4 + x5, yd = x4 - x5;
    g.y2(ys, yd);
    x2 = mad24(181, ys, 128) >> 8;
    x4 = mad24(181, yd, 128) >> 8;
    // fourth stage
    constexpr int S = COL ? 0 : 8;
    b0 = (x7 + x1) >> S;
    b1 = (x3 + x2) >> S;
    b2 = (x0 + x4) >> S;
    b3 = (x8 + x6) >> S;
    b4 = (x8 - x6) >> S;
    b5 = (x0 - x4) >> S;
    b6 = (x3 - x2) >> S;
    b7 = (x7 - x1) >> S;
}

// gfx950 V_ASHR_PK_U8_I32: D.u16 = { sat_u8(S1 >>> n), sat_u8(S0 >>> n) } -- a
// 16-bit write: op_sel[3] selects the destination half and the OTHER HALF IS
// PRESERVED.  (hipcc 7.2 pattern-matches `clamp(x >> n, 0, 255)` pairs into this
// instruction but then treats the result as zero-extended, which is wrong on
// hardware; so the kernel never leaves that pattern to the compiler and issues
// the instruction itself, once per destination half: 4 pixels = 2 instructions
// for shift + clamp + pack.)
__device__ __forceinline__ unsigned ashr14_sat_pack4(int a, int b, int c, int d) {
    unsigned r;
    asm("v_ashr_pk_u8_i32 %0, %1, %2, 14" : "=v"(r) : "v"(a), "v"(b));
    asm("v_ashr_pk_u8_i32 %0, %1, %2, 14 op_sel:[0,0,0,1]" : "+v"(r) : "v"(c), "v"(d));
    return r;
}

// The same pass in int64, literally as the model writes it (dct.ml:11-98).
template <bool COL>
__device__ __forceinline__ void idct_1d_wide(int64_t *b, int s) {
    int64_t x0 = COL ? b[0] * 256 + 8192 : b[0] * 2048 + 128;
    int64_t x1 = COL ? b[4 * s] * 256 : b[4 * s] * 2048;
    int64_t x2 = b[6 * s], x3 = b[2 * s], x4 = b[1 * s], x5 = b[7 * s], x6 = b[5 * s], x7 = b[3 * s];
    constexpr int64_t R = COL ? 4 : 0;
    constexpr int RS = COL ? 3 : 0;
    int64_t x8 = W7 * (x4 + x5) + R;
    x4 = (x8 + (W1 - W7) * x4) >> RS;
    x5 = (x8 - (W1 + W7) * x5) >> RS;
    x8 = W3 * (x6 + x7) + R;
    x6 = (x8 - (W3 - W5) * x6) >> RS;
    x7 = (x8 - (W3 + W5) * x7) >> RS;
    x8 = x0 + x1;
    x0 = x0 - x1;
    x1 = W6 * (x3 + x2) + R;
    x2 = (x1 - (W2 + W6) * x2) >> RS;
    x3 = (x1 + (W2 - W6) * x3) >> RS;
    x1 = x4 + x6;
    x4 = x4 - x6;
    x6 = x5 + x7;
    x5 = x5 - x7;
    x7 = x8 + x3;
    x8 = x8 - x3;
    x3 = x0 + x2;
    x0 = x0 - x2;
    int64_t ys = x4 + x5, yd = x4 - x5;
    x2 = (181 * ys + 128) >> 8;
    x4 = (181 * yd + 128) >> 8;
    constexpr int S = COL ? 14 : 8;
    b[0] = (x7 + x1) >> S;
    b[1 * s] = (x3 + x2) >> S;
    b[2 * s] = (x0 + x4) >> S;
    b[3 * s] = (x8 + x6) >> S;
    b[4 * s] = (x8 - x6) >> S;
    b[5 * s] = (x0 - x4) >> S;
    b[6 * s] = (x3 - x2) >> S;
    b[7 * s] = (x7 - x1) >> S;
}

// ---------------------------------------------------------------------------
// Work decomposition shared by the fast and the wide kernel.
// grid.x = tiles per frame (a tile = 256 consecutive blocks of one component
// plane), grid.y = frame.  Returns false for lanes past the end of the plane.
struct BlockRef {
    size_t coef_idx; // int16 element index of this block's 64 coefficients
    size_t pix_idx;  // byte index of this block's top-left pixel
    size_t stride;
    int qtab;
};

template <class Params>
__device__ __forceinline__ bool locate(const Params &P, int frame, int tile, int lane, BlockRef &br) {
    int c = 0;
#pragma unroll
    for (int i = 1; i < HVC_MAX_COMP; i++)
        if (i < P.n_comp && tile >= P.comp[i].tile0) c = i;
    const CompK &K = P.comp[c];
    int b = (tile - K.tile0) * HVC_TILE + lane;
    bool active = b < K.nblk;
    b = active ? b : K.nblk - 1;
    unsigned by = K.bw == 1 ? (unsigned)b : __umulhi((unsigned)b, K.magic);
    unsigned bx = (unsigned)b - by * (unsigned)K.bw;
    br.coef_idx = (size_t)frame * P.coef_fs + K.coef_off + (size_t)b * 64;
    br.pix_idx = (size_t)frame * P.pixel_fs + K.plane_off + (size_t)by * 8 * K.stride + (size_t)bx * 8;
    br.stride = K.stride;
    br.qtab = K.qtab;
    return active;
}

// ---------------------------------------------------------------------------
// K1 fast: int32 / mul24.
__global__ __launch_bounds__(HVC_FAST_LB) void k_decode_fast(DecodeParams P) {
    BlockRef br;
    const int lane = threadIdx.x;
    const bool active = locate(P, blockIdx.y, blockIdx.x, lane, br);

    // 8 x 16 B per lane: the wave's loads cover one contiguous 8 KiB run.
    const uint4 *src = reinterpret_cast<const uint4 *>(P.coefs + br.coef_idx);
    uint4 raw[8];
#pragma unroll
    for (int j = 0; j < 8; j++) raw[j] = src[j];

    const int *__restrict__ q = P.qt + br.qtab * 64; // wave-uniform, kernarg segment -> scalar loads

    // dequantise + inverse zig-zag (decoder.ml:142-149): v[ZI[k]] = c[k] * q[k]
    int v[64];
    Guard g;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const unsigned w[4] = {raw[j].x, raw[j].y, raw[j].z, raw[j].w};
#pragma unroll
        for (int h = 0; h < 4; h++) {
            const int k = j * 8 + h * 2;
            g.e2(w[h]);
            v[ZI[k]] = dequant_lo(w[h], q[k]);
            v[ZI[k + 1]] = dequant_hi(w[h], q[k + 1]);
        }
    }

    // Dct.Chen.inverse_8x8 (dct.ml:100-107): rows, then columns
#pragma unroll
    for (int r = 0; r < 8; r++) {
        idct_1d_fast<false, 0>(v[r * 8 + 0], v[r * 8 + 1], v[r * 8 + 2], v[r * 8 + 3], v[r * 8 + 4],
                               v[r * 8 + 5], v[r * 8 + 6], v[r * 8 + 7], g);
        g.r2(v[r * 8 + 0], v[r * 8 + 1]);
        g.r2(v[r * 8 + 2], v[r * 8 + 3]);
        g.r2(v[r * 8 + 4], v[r * 8 + 5]);
        g.r2(v[r * 8 + 6], v[r * 8 + 7]);
    }
    // the +128 level shift of recon (decoder.ml:220) is folded into the column
    // pass: ((s + (128 << 14)) >> 14) == (s >> 14) + 128 exactly.
#pragma unroll
    for (int c = 0; c < 8; c++)
        idct_1d_fast<true, (128 << 14)>(v[c], v[8 + c], v[16 + c], v[24 + c], v[32 + c], v[40 + c],
                                        v[48 + c], v[56 + c], g);

    // clip (decoder.ml:213) on the shifted value: clamp(x,-128,127)+128 == clamp(x+128,0,255),
    // done together with the column pass's >> 14 by the saturating pack.
    const bool bad = g.bad(P.ethr[br.qtab]);
    if (active && !bad) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            uint2 o;
            o.x = ashr14_sat_pack4(v[j * 8 + 0], v[j * 8 + 1], v[j * 8 + 2], v[j * 8 + 3]);
            o.y = ashr14_sat_pack4(v[j * 8 + 4], v[j * 8 + 5], v[j * 8 + 6], v[j * 8 + 7]);
            *reinterpret_cast<uint2 *>(P.pixels + br.pix_idx + (size_t)j * br.stride) = o;
        }
    }
    // fix-up list: one atomic per wave
    const bool flag = active && bad;
    const unsigned long long m = __ballot(flag);
    if (m) {
        const int wl = lane & 63;
        unsigned base = 0;
        if (wl == 0) base = atomicAdd(P.fix_count, (unsigned)__popcll(m));
        base = __shfl(base, 0);
        if (flag) {
            unsigned idx = base + (unsigned)__popcll(m & ((1ull << wl) - 1ull));
            P.fix_list[idx] = ((unsigned)blockIdx.y * (unsigned)P.tiles_per_frame + blockIdx.x) * HVC_TILE + lane;
        }
    }
}

// K1 wide: int64, one flagged block per thread; also usable on its own for a
// whole batch (list == nullptr: every block).
__global__ __launch_bounds__(64) void k_decode_wide(DecodeParams P, const unsigned *count, const unsigned *list,
                                                    unsigned long long total) {
    unsigned long long n = list ? (unsigned long long)*count : total;
    // The two fix-up counters alternate between calls: this launch reads the
    // current one and clears the other for the next call (no memset node).
    if (blockIdx.x == 0 && threadIdx.x == 0 && P.fix_count_next) *P.fix_count_next = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * 64 + threadIdx.x; i < n;
         i += (unsigned long long)gridDim.x * 64) {
        unsigned long long id = list ? list[i] : i;
        int lane = (int)(id % HVC_TILE);
        unsigned long long t = id / HVC_TILE;
        int tile = (int)(t % (unsigned)P.tiles_per_frame);
        int frame = (int)(t / (unsigned)P.tiles_per_frame);
        BlockRef br;
        if (!locate(P, frame, tile, lane, br)) continue;
        const int *q = P.qt + br.qtab * 64;
        int64_t v[64];
        const int16_t *cf = P.coefs + br.coef_idx;
        for (int k = 0; k < 64; k++) v[ZI[k]] = (int64_t)cf[k] * (int64_t)q[k];
        for (int r = 0; r < 8; r++) idct_1d_wide<false>(v + r * 8, 1);
        for (int c = 0; c < 8; c++) idct_1d_wide<true>(v + c, 8);
        for (int j = 0; j < 8; j++)
            for (int i2 = 0; i2 < 8; i2++) {
                int64_t x = v[j * 8 + i2];
                x = x < -128 ? -128 : (x > 127 ? 127 : x);
                P.pixels[br.pix_idx + (size_t)j * br.stride + i2] = (uint8_t)(x + 128);
            }
    }
}

// ---------------------------------------------------------------------------
// K3: level shift + Dct.Chen.forward_8x8 + quantise + zig-zag
//   jpeg/model/src/encoder.ml:81-108, jpeg/model/src/dct.ml:109-196.
// Inputs are 8-bit pixels, so every intermediate is bounded a priori (|p-128| <=
// 128, first pass <= 2^11, second pass <= 2^14; tests/test_guard_bounds.py): int32
// with 24-bit multiplies is exact, no guard and no wide kernel.

// dct.ml:109-112
__device__ __forceinline__ int c4(int f, int g) { return mul24(362, f + g) >> 9; }
__device__ __forceinline__ int c4m(int f, int g) { return mul24(362, f - g) >> 9; }
__device__ __forceinline__ int c62(int f, int g) { return mad24(473, g, mul24(196, f)) >> 9; }
__device__ __forceinline__ int c71(int f, int g) { return mad24(502, g, mul24(100, f)) >> 9; }
__device__ __forceinline__ int c35(int f, int g) { return mad24(284, g, mul24(426, f)) >> 9; }

// dct.ml:114-149 (dct_col) / :151-187 (dct_row): one butterfly, 8 values in place
__device__ __forceinline__ void fdct_1d(int &p0, int &p1, int &p2, int &p3, int &p4, int &p5, int &p6, int &p7) {
    int a0 = p0 + p7, c3 = p0 - p7;
    int a1 = p1 + p6, c2 = p1 - p6;
    int a2 = p2 + p5, c1 = p2 - p5;
    int a3 = p3 + p4, c0 = p3 - p4;
    int b0 = a0 + a3, b1 = a1 + a2, b2 = a1 - a2, b3 = a0 - a3;
    p0 = c4(b0, b1);
    p4 = c4m(b0, b1);      // c4 b0 (-b1)
    p2 = c62(b2, b3);
    p6 = c62(b3, -b2);
    b0 = c4m(c2, c1);      // c4 c2 (-c1)
    b1 = c4(c2, c1);
    a0 = c0 + b0;
    a1 = c0 - b0;
    a2 = c3 - b1;
    a3 = c3 + b1;
    p1 = c71(a0, a3);
    p5 = c35(a1, a2);
    p3 = c35(a2, -a1);
    p7 = c71(a3, -a0);
}

// Encoder.quant_and_scale (encoder.ml:98-101): trunc((f +- 2t) / (4t)), i.e.
// f/(4t) rounded half away from zero.  Evaluated as trunc(|f| * r + h) with
// r = fl(1/(4t)) and h = 0.5 + 1/(8t): the exact quotients are multiples of
// 1/(4t), the bias centres them between float errors (< 2^-9 / (4t) for
// |f| < 2^14), so the truncation is exact -- verified exhaustively for every
// t in 1..255 and every |f| <= 2^15 by tests/test_quant_division.py.
__device__ __forceinline__ int quant1(int f, float r, float h) {
    float ff = (float)f;
    float x = __builtin_fmaf(__builtin_fabsf(ff), r, h);
    x = __builtin_copysignf(x, ff);
    return (int)x; // v_cvt_i32_f32 truncates toward zero
}

__global__ __launch_bounds__(HVC_TILE) void k_encode(EncodeParams P) {
    BlockRef br;
    const int lane = threadIdx.x;
    const bool active = locate(P, blockIdx.y, blockIdx.x, lane, br);
    const uint8_t *pix = P.pixels + br.pix_idx;
    int v[64];
    // level_shifted_input_block (encoder.ml:81-90): 8 rows x 8 B per lane; a
    // wave's row loads are 512 contiguous bytes of a pixel row.
#pragma unroll
    for (int j = 0; j < 8; j++) {
        uint2 w = *reinterpret_cast<const uint2 *>(pix + (size_t)j * br.stride);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            v[j * 8 + i] = (int)((w.x >> (8 * i)) & 0xffu) - 128;
            v[j * 8 + 4 + i] = (int)((w.y >> (8 * i)) & 0xffu) - 128;
        }
    }
    // Dct.Chen.forward_8x8 (dct.ml:189-196): columns first, then rows
#pragma unroll
    for (int c = 0; c < 8; c++)
        fdct_1d(v[c], v[8 + c], v[16 + c], v[24 + c], v[32 + c], v[40 + c], v[48 + c], v[56 + c]);
#pragma unroll
    for (int r = 0; r < 8; r++)
        fdct_1d(v[r * 8 + 0], v[r * 8 + 1], v[r * 8 + 2], v[r * 8 + 3], v[r * 8 + 4], v[r * 8 + 5], v[r * 8 + 6],
                v[r * 8 + 7]);
    // Encoder.quant (encoder.ml:103-108): quant[zz] = quant_and_scale fdct[ZI[zz]] table[zz]
    const float *__restrict__ qr = P.qrcp + br.qtab * 64;
    const float *__restrict__ qh = P.qhalf + br.qtab * 64;
    uint4 *dst = reinterpret_cast<uint4 *>(P.coefs + br.coef_idx);
#pragma unroll
    for (int j = 0; j < 8; j++) {
        unsigned w[4];
#pragma unroll
        for (int h = 0; h < 4; h++) {
            const int k = j * 8 + h * 2;
            int lo = quant1(v[ZI[k]], qr[k], qh[k]);
            int hi = quant1(v[ZI[k + 1]], qr[k + 1], qh[k + 1]);
            w[h] = ((unsigned)lo & 0xffffu) | ((unsigned)hi << 16);
        }
        if (active) dst[j] = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// ---------------------------------------------------------------------------
// K2: 4:2:0 -> 4:4:4 chroma upsample, tools/src/planar_444.ml:82-103
// (supersample_hv2 for every row, :122-131).  One thread per 4 source pixels of
// a source row: writes 8 + 8 destination bytes (rows 2r and 2r+1).
__device__ __forceinline__ unsigned avg2u(unsigned a, unsigned b) { return (a + b + 1) >> 1; }          // :4-8
__device__ __forceinline__ unsigned avg4u(unsigned a, unsigned b, unsigned c, unsigned d) { return (a + b + c + d + 2) >> 2; } // :10-16

__global__ __launch_bounds__(256) void k_upsample420(UpsampleParams P) {
    const int groups = (P.cw + 3) >> 2;
    const long long total = (long long)groups * P.ch;
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int row = (int)(t / groups), g = (int)(t % groups);
    const int plane = blockIdx.y;
    const uint8_t *src = P.src + (size_t)plane * P.src_ps;
    uint8_t *dst = P.dst + (size_t)plane * P.dst_ps;
    const int row2 = min(P.ch - 1, row + 1);
    const uint8_t *s1 = src + (size_t)row * P.src_stride;
    const uint8_t *s2 = src + (size_t)row2 * P.src_stride;
    const int c0 = g * 4;
    unsigned a[5], b[5];
#pragma unroll
    for (int i = 0; i < 5; i++) {
        int c = min(c0 + i, P.cw - 1); // last column replicates (:98-102)
        a[i] = s1[c];
        b[i] = s2[c];
    }
    uint8_t *d1 = dst + (size_t)(2 * row) * P.dst_stride + 2 * c0;
    uint8_t *d2 = d1 + P.dst_stride;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        if (c0 + i >= P.cw) break;
        d1[2 * i] = (uint8_t)a[i];
        d1[2 * i + 1] = (uint8_t)avg2u(a[i], a[i + 1]);
        d2[2 * i] = (uint8_t)avg2u(a[i], b[i]);
        d2[2 * i + 1] = (uint8_t)avg4u(a[i], a[i + 1], b[i], b[i + 1]);
    }
}

// ---------------------------------------------------------------------------
// launchers (host)
hipError_t launch_decode(const DecodeParams &P, hipStream_t s, hipEvent_t k0, hipEvent_t k1) {
    if (P.n_frames <= 0 || P.tiles_per_frame <= 0) return hipSuccess;
    hipError_t e;
    dim3 grid((unsigned)P.tiles_per_frame, (unsigned)P.n_frames, 1);
    if (k0 && (e = hipEventRecord(k0, s)) != hipSuccess) return e;
    hipLaunchKernelGGL(k_decode_fast, grid, dim3(HVC_TILE), 0, s, P);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (k1 && (e = hipEventRecord(k1, s)) != hipSuccess) return e;
    // Fixed small grid; every thread strides over the (normally empty) list and
    // exits as soon as its index passes *fix_count.
    hipLaunchKernelGGL(k_decode_wide, dim3(256), dim3(64), 0, s, P, P.fix_count, P.fix_list, 0ull);
    return hipGetLastError();
}

hipError_t launch_decode_wide_only(const DecodeParams &P, hipStream_t s) {
    if (P.n_frames <= 0 || P.tiles_per_frame <= 0) return hipSuccess;
    unsigned long long total = (unsigned long long)P.n_frames * P.tiles_per_frame * HVC_TILE;
    hipLaunchKernelGGL(k_decode_wide, dim3(4096), dim3(64), 0, s, P, (const unsigned *)nullptr,
                       (const unsigned *)nullptr, total);
    return hipGetLastError();
}

hipError_t launch_encode(const EncodeParams &P, hipStream_t s, hipEvent_t k0, hipEvent_t k1) {
    if (P.n_frames <= 0 || P.tiles_per_frame <= 0) return hipSuccess;
    hipError_t e;
    dim3 grid((unsigned)P.tiles_per_frame, (unsigned)P.n_frames, 1);
    if (k0 && (e = hipEventRecord(k0, s)) != hipSuccess) return e;
    hipLaunchKernelGGL(k_encode, grid, dim3(HVC_TILE), 0, s, P);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (k1 && (e = hipEventRecord(k1, s)) != hipSuccess) return e;
    return hipSuccess;
}

hipError_t launch_upsample420(const UpsampleParams &P, hipStream_t s) {
    if (P.n_planes <= 0 || P.cw <= 0 || P.ch <= 0) return hipSuccess;
    long long total = (long long)((P.cw + 3) >> 2) * P.ch;
    dim3 grid((unsigned)((total + 255) / 256), (unsigned)P.n_planes, 1);
    hipLaunchKernelGGL(k_upsample420, grid, dim3(256), 0, s, P);
    return hipGetLastError();
}

} // namespace hvc
