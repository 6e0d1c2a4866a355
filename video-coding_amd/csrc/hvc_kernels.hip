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
#include <stdlib.h>
#include <string.h>

#include "hvc_idct_spec.h"
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
constexpr int W1 = HVC_W1, W2 = HVC_W2, W3 = HVC_W3, W5 = HVC_W5, W6 = HVC_W6, W7 = HVC_W7;

// ---------------------------------------------------------------------------
// Range guard of the int32 fast path (proved by tests/test_guard_bounds.py with
// interval arithmetic over exactly the operations below):
//   |dequantised coefficient|        <= GUARD_D   (via the coefficient energy, below)
//   |row-pass output|                <= GUARD_R
//   |argument of a 181*y product|    <= GUARD_Y   (row and column pass)
// imply that no int32 operation wraps and every v_mul_i32_i24 / v_mad_i32_i24
// operand lies in [-2^23, 2^23).
[[maybe_unused]] constexpr int GUARD_D = HVC_GUARD_D;
constexpr int GUARD_R = (1 << 18) - 1;
constexpr int GUARD_Y = HVC_GUARD_Y;

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
// 8 pixels of one block row.  The pixels are written once and never re-read by this kernel: a
// non-temporal store (HVC_NT_STORES, default) measured +2.6 % on the kernel's traffic shape
// (profiles/r01_mem_ubench.txt: 5.76 -> 5.91 TB/s); non-temporal LOADS would halve it (they bypass
// the L1 that merges a lane's eight 16-byte reads of its 128-byte line).
#ifndef HVC_NT_STORES
#define HVC_NT_STORES 1
#endif
__device__ __forceinline__ void store_row8(uint8_t *p, unsigned lo, unsigned hi) {
    typedef unsigned u2v __attribute__((ext_vector_type(2)));
    u2v t = {lo, hi};
#if HVC_NT_STORES
    __builtin_nontemporal_store(t, reinterpret_cast<u2v *>(p));
#else
    *reinterpret_cast<u2v *>(p) = t;
#endif
}

__device__ __forceinline__ unsigned ashr14_sat_pack4(int a, int b, int c, int d) {
    unsigned r;
    asm("v_ashr_pk_u8_i32 %0, %1, %2, 14" : "=v"(r) : "v"(a), "v"(b));
    asm("v_ashr_pk_u8_i32 %0, %1, %2, 14 op_sel:[0,0,0,1]" : "+v"(r) : "v"(c), "v"(d));
    return r;
}

// The same pass as the model writes it (dct.ml:11-98), in the model's arithmetic: OCaml's int is a 63-bit two's
// complement number that wraps without a word.  Sums and products are kept modulo 2^64 (unsigned: congruent modulo
// 2^63 to the model's values), and a value is brought to its 63-bit reading where the model LOOKS at it: the operand of
// an `asr`, the compare of clip.  For everything 16-bit coefficients and 16-bit tables can reach (< 2^47) the reading is
// the identity; it matters for the DC values a table with 33...62-bit categories produces (include/hvc_jpeg.h).
__device__ __forceinline__ int64_t ocaml_int(uint64_t x) { return (int64_t)(x << 1) >> 1; }
// WRAP63 = false: the caller knows every value of the pass stays below 2^62 in magnitude, where the 63-bit reading is the
// identity and `asr` is one 64-bit shift.  That is every block whose 64 inputs are an int16 coefficient times a 16-bit table
// entry (|input| < 2^31: no value of either pass reaches 2^57 -- replayed on
// magnitudes by tests/test_guard_bounds.py::test_wide_kernel_without_the_63_bit_reading); only a DC from the side list of
// DCs beyond int16 (a 33 ... 62-bit DC category) can reach the wrap-around, and those blocks go through the WRAP63 form.
template <int S, bool WRAP63>
__device__ __forceinline__ uint64_t asr63(uint64_t x) {
    return !S ? x : WRAP63 ? (uint64_t)(ocaml_int(x) >> S) : (uint64_t)((int64_t)x >> S);
}

// Eight values in, eight out, all in registers: b[i] is the pass's input i (a row's or a column's i-th element).
template <bool COL, bool WRAP63>
__device__ __forceinline__ void idct8_wide(const uint64_t (&b)[8], int64_t (&o)[8]) {
    typedef uint64_t u64;
    constexpr u64 w1 = (u64)W1, w2 = (u64)W2, w3 = (u64)W3, w5 = (u64)W5, w6 = (u64)W6, w7 = (u64)W7;
    u64 x0 = COL ? b[0] * 256u + 8192u : b[0] * 2048u + 128u;
    u64 x1 = COL ? b[4] * 256u : b[4] * 2048u;
    u64 x2 = b[6], x3 = b[2], x4 = b[1], x5 = b[7], x6 = b[5], x7 = b[3];
    constexpr u64 R = COL ? 4 : 0;
    constexpr int RS = COL ? 3 : 0;
    u64 x8 = w7 * (x4 + x5) + R;
    x4 = asr63<RS, WRAP63>(x8 + (w1 - w7) * x4);
    x5 = asr63<RS, WRAP63>(x8 - (w1 + w7) * x5);
    x8 = w3 * (x6 + x7) + R;
    x6 = asr63<RS, WRAP63>(x8 - (w3 - w5) * x6);
    x7 = asr63<RS, WRAP63>(x8 - (w3 + w5) * x7);
    x8 = x0 + x1;
    x0 = x0 - x1;
    x1 = w6 * (x3 + x2) + R;
    x2 = asr63<RS, WRAP63>(x1 - (w2 + w6) * x2);
    x3 = asr63<RS, WRAP63>(x1 + (w2 - w6) * x3);
    x1 = x4 + x6;
    x4 = x4 - x6;
    x6 = x5 + x7;
    x5 = x5 - x7;
    x7 = x8 + x3;
    x8 = x8 - x3;
    x3 = x0 + x2;
    x0 = x0 - x2;
    const u64 ys = x4 + x5, yd = x4 - x5;
    x2 = asr63<8, WRAP63>(181u * ys + 128u);
    x4 = asr63<8, WRAP63>(181u * yd + 128u);
    constexpr int S = COL ? 14 : 8;
    o[0] = (int64_t)asr63<S, WRAP63>(x7 + x1);
    o[1] = (int64_t)asr63<S, WRAP63>(x3 + x2);
    o[2] = (int64_t)asr63<S, WRAP63>(x0 + x4);
    o[3] = (int64_t)asr63<S, WRAP63>(x8 + x6);
    o[4] = (int64_t)asr63<S, WRAP63>(x8 - x6);
    o[5] = (int64_t)asr63<S, WRAP63>(x0 - x4);
    o[6] = (int64_t)asr63<S, WRAP63>(x3 - x2);
    o[7] = (int64_t)asr63<S, WRAP63>(x7 - x1);
}
// decoder.ml:142-149 `coefs.(i) * qnt_tab.(i)` in the same arithmetic (the factors: an int16 or a 63-bit DC, a 16-bit entry)
__device__ __forceinline__ int64_t mul63(int64_t a, int64_t b) { return (int64_t)((uint64_t)a * (uint64_t)b); }

// ---------------------------------------------------------------------------
// Work decomposition shared by the fast and the wide kernel.
// grid.x = tiles per frame (a tile = 256 consecutive blocks of one component
// plane), grid.y = frame.  Returns false for lanes past the end of the plane.
struct BlockRef {
    size_t coef_idx; // int16 element index of this block's 64 coefficients
    size_t pix_idx;  // byte index of this block's top-left pixel
    size_t stride;
    int qtab;
    size_t plane_coef_idx; // int16 element index of the plane's first coefficient
    int tile_b0;           // block index (inside the plane) of lane 0 of this workgroup
    int nblk;              // blocks in the plane
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
    br.plane_coef_idx = (size_t)frame * P.coef_fs + K.coef_off;
    br.tile_b0 = (tile - K.tile0) * HVC_TILE;
    br.nblk = K.nblk;
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
            store_row8(P.pixels + br.pix_idx + (size_t)j * br.stride, o.x, o.y);
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

// ---------------------------------------------------------------------------
// K1 packed: the same block-per-lane mapping with PACKED int16 operands.
//
// Measured issue costs (profiles/r01_valu_ubench.txt) make integer multiplies the
// expensive part of the butterfly (mul/mad ~1.7x an add).  v_dot2_i32_i16
// (d = a.lo*b.lo + a.hi*b.hi + c, same ~1.7x cost) computes one whole rotation
// output  W1*x4 + W7*x5 (+ rounding)  per instruction when (x4, x5) sit in one
// register as an int16 pair, and the pair x8 = (b0 + b4) << s + r,
// x0 - x1 = (b0 - b4) << s + r the same way.  So:
//   * the coefficient dwords are byte-permuted (v_perm_b32) into the four
//     operand pairs of each row -- (b1,b7) (b5,b3) (b2,b6) (b0,b4) -- and
//     dequantised two at a time (v_pk_mul_lo_u16, exact while |c*q| < 2^15);
//   * a row pass is 8 v_dot2 + adds; its outputs are saturate-packed
//     (v_cvt_pk_i16_i32) across ROWS into the column passes' operand pairs
//     (r1,r7) (r5,r3) (r2,r6) (r0,r4): 32 VGPRs instead of 64 for the transposed
//     intermediate;
//   * a column pass is again 8 v_dot2 + adds, finished two columns at a time by
//     v_ashr_pk_u8_i32 (>> 14, clip, byte pack).
// Guard (tests/test_guard_bounds.py): coefficient energy E <= (32767/qmax)^2 =>
// every |c*q| <= 32767; row-output energy sum r^2 < 32767^2 computed on the
// SATURATED packed values (a clipped value alone reaches the threshold) => every
// |r| < 32767 and the pack was exact; |y| < 2^23 for the two 181*y products of
// every pass.  Anything else goes to the int64 kernel via the fix-up list.
typedef short short2v __attribute__((ext_vector_type(2)));
typedef unsigned short ushort2v __attribute__((ext_vector_type(2)));

// d = pair.lo * k.lo + pair.hi * k.hi + ADD with the three-operand (VOP3P) encoding: the constant pair
// sits in an SGPR and a small addend is an inline constant, so no v_mov is spent on the accumulator
// (the builtin lowers to the two-operand v_dot2c form, which needs d preloaded).
template <int ADD>
__device__ __forceinline__ int dot2(unsigned pair, unsigned k) {
    static_assert(ADD == 0 || ADD == 4, "inline constants only");
    int d;
    if (ADD == 0)
        asm("v_dot2_i32_i16 %0, %1, %2, 0" : "=v"(d) : "v"(pair), "s"(k));
    else
        asm("v_dot2_i32_i16 %0, %1, %2, 4" : "=v"(d) : "v"(pair), "s"(k));
    return d;
}
// same with a (wave-uniform) addend that is not an inline constant: it lives in a VGPR
__device__ __forceinline__ int dot2v(unsigned pair, unsigned k, int add) {
    int d;
    asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(d) : "v"(pair), "s"(k), "v"(add));
    return d;
}
__device__ __forceinline__ int dot2_sat(unsigned pair, unsigned k, int acc) {
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, pair), __builtin_bit_cast(short2v, k), acc, true);
}
constexpr unsigned pk(int lo, int hi) { return ((unsigned)lo & 0xffffu) | ((unsigned)hi << 16); }

// position of raster coefficient p inside the 32 loaded dwords: dword ZF[p] / 2, half ZF[p] & 1
template <int PA, int PB>
__device__ __forceinline__ unsigned gather_pair(const unsigned (&w)[32]) {
    constexpr int za = ZF[PA], zb = ZF[PB];
    constexpr int ha = za & 1, hb = zb & 1;
    // v_perm_b32 D = bytes of {S0 (bytes 4-7), S1 (bytes 0-3)} chosen by the selector
    constexpr unsigned sel = ((unsigned)(4 + 2 * hb + 1) << 24) | ((unsigned)(4 + 2 * hb) << 16) |
                             ((unsigned)(2 * ha + 1) << 8) | (unsigned)(2 * ha);
    return __builtin_amdgcn_perm(w[zb >> 1], w[za >> 1], sel);
}
__device__ __forceinline__ unsigned pk_mul_lo(unsigned a, unsigned b) {
    return __builtin_bit_cast(unsigned, (ushort2v)(__builtin_bit_cast(ushort2v, a) * __builtin_bit_cast(ushort2v, b)));
}

struct PackedGuard {
    int energy = 0;   // sum of squared quantised coefficients (saturating)
    int renergy = 0;  // sum of squared (saturated) row outputs (saturating)
    int ymax = 0, ymin = 0;
    int k128 = HVC_ROW_ZADD, kcol = HVC_COL_ZADD; // wave-uniform addends of the (b0, b4) dot products, kept in VGPRs
    __device__ __forceinline__ void y2(int a, int b) { ymax = max(max(ymax, a), b); ymin = min(min(ymin, a), b); }
};
constexpr int GUARD_RE = HVC_GUARD_RE; // renergy >= this  <=>  some |r| may have reached 32767

// The statements of a pass are the expansion of HVC_IDCT_PASS (hvc_idct_spec.h): the operation list that
// tests/test_guard_bounds.py replays on intervals.  PASS = ROW or COL picks the parameter set (the wave-uniform
// addend of the (b0, b4) dot products sits in a VGPR of the guard struct: g.k128 / g.kcol).
#define HVC_EXPAND_PASS(PASS)                                                                                           \
    HVC_IDCT_PASS(HVC_OP_ROT_##PASS, HVC_OP_ZDOT_##PASS, HVC_OP_ADD, HVC_OP_SUB, HVC_OP_GUARDY, HVC_OP_M181,            \
                  HVC_OP_OUTADD_##PASS, HVC_OP_OUTSUB_##PASS)
#define HVC_OP_ROT_ROW(d, P, klo, khi) const int d = dot2<HVC_ROW_RADD>(P, pk(klo, khi)) >> HVC_ROW_RSHIFT;
#define HVC_OP_ROT_COL(d, P, klo, khi) const int d = dot2<HVC_COL_RADD>(P, pk(klo, khi)) >> HVC_COL_RSHIFT;
#define HVC_OP_ZDOT_ROW(d, P, slo, shi) const int d = dot2v(P, pk((slo) * HVC_ROW_ZSCALE, (shi) * HVC_ROW_ZSCALE), g.k128);
#define HVC_OP_ZDOT_COL(d, P, slo, shi) const int d = dot2v(P, pk((slo) * HVC_COL_ZSCALE, (shi) * HVC_COL_ZSCALE), g.kcol);
#define HVC_OP_ADD(d, a, b) const int d = a + b;
#define HVC_OP_SUB(d, a, b) const int d = a - b;
#define HVC_OP_GUARDY(a, b) g.y2(a, b);
#define HVC_OP_M181(d, a) const int d = mad24(HVC_M181_MUL, a, HVC_M181_ADD) >> HVC_M181_SHIFT;
#define HVC_OP_OUTADD_ROW(i, a, b) o[i] = (a + b) >> HVC_ROW_OSHIFT;
#define HVC_OP_OUTSUB_ROW(i, a, b) o[i] = (a - b) >> HVC_ROW_OSHIFT;
#define HVC_OP_OUTADD_COL(i, a, b) o[i] = (a + b) >> HVC_COL_OSHIFT;
#define HVC_OP_OUTSUB_COL(i, a, b) o[i] = (a - b) >> HVC_COL_OSHIFT;

// Row pass (dct.ml:11-54) of row R from the coefficient dwords; the eight outputs (>> 8) are returned
// in o[0..7].  qp = the row's four packed quantiser pairs, in the order A, B, C, Z of hvc_idct_spec.h.
template <int R>
__device__ __forceinline__ void idct_row_packed(const unsigned (&w)[32], const unsigned *__restrict__ qp, int (&o)[8],
                                                PackedGuard &g) {
    const unsigned A = pk_mul_lo(gather_pair<8 * R + HVC_PAIR_A_LO, 8 * R + HVC_PAIR_A_HI>(w), qp[0]); // (x4, x5)
    const unsigned B = pk_mul_lo(gather_pair<8 * R + HVC_PAIR_B_LO, 8 * R + HVC_PAIR_B_HI>(w), qp[1]); // (x6, x7)
    const unsigned C = pk_mul_lo(gather_pair<8 * R + HVC_PAIR_C_LO, 8 * R + HVC_PAIR_C_HI>(w), qp[2]); // (x3, x2)
    const unsigned Z = pk_mul_lo(gather_pair<8 * R + HVC_PAIR_Z_LO, 8 * R + HVC_PAIR_Z_HI>(w), qp[3]); // (b0, b4)
    HVC_EXPAND_PASS(ROW)
}

// saturating pack of two row outputs into one column-pass operand pair + its share of the energy
__device__ __forceinline__ unsigned pack_rows(int lo, int hi, PackedGuard &g) {
    const unsigned p = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pk_i16(lo, hi));
    g.renergy = dot2_sat(p, p, g.renergy);
    return p;
}

// Column pass (dct.ml:56-98) from the four operand pairs of one column; outputs unshifted, with the
// +128 level shift of recon folded into the rounding constant (see k_decode_fast, hvc_idct_spec.h).
__device__ __forceinline__ void idct_col_packed(unsigned A, unsigned B, unsigned C, unsigned Z, int (&o)[8],
                                                PackedGuard &g) {
    HVC_EXPAND_PASS(COL)
}

// two adjacent pixels of a row: sat_u8(a >> 14) | sat_u8(b >> 14) << 8 into one half of dst
template <int HALF>
__device__ __forceinline__ void ashr14_sat_pack2(unsigned &dst, int a, int b) {
    if (HALF == 0)
        asm("v_ashr_pk_u8_i32 %0, %1, %2, %3" : "=v"(dst) : "v"(a), "v"(b), "n"(HVC_COL_PACK_SHIFT));
    else
        asm("v_ashr_pk_u8_i32 %0, %1, %2, %3 op_sel:[0,0,0,1]" : "+v"(dst) : "v"(a), "v"(b), "n"(HVC_COL_PACK_SHIFT));
}

#ifndef HVC_PACKED_LB
// Occupancy A/B on MI355X (same box, 40 steps, 256 frames): asking for 7 waves/SIMD makes the
// scheduler stay within 72 VGPRs (66, no spill) and runs 0.453 ms; 5-6 waves 0.436-0.438 ms; left
// free (<= 4 waves: 116 VGPRs, rows interleaved for ILP) 0.432 ms; 8 waves needs spills, 0.476 ms.
// The kernel is HBM-bound (mem_ubench ceiling for this traffic shape 0.407 ms), so fewer, longer
// waves with bursty memory phases win.
#define HVC_PACKED_LB HVC_TILE, 4
#endif
// One block per lane: 128 B of coefficients (SRC: const uint4 *) -> the block's 8 pixel rows as
// byte-packed dword pairs OUT[8][2]; QP = the table's 32 packed quantiser pairs; G = PackedGuard.
// A macro, not a function: spelled inline, hipcc 7.2 schedules the kernel for <= 4 waves/SIMD
// (116 VGPRs, rows interleaved); through an (always-inlined) function it settles on 64 VGPRs /
// 8 waves, measured 6 % slower on the same box (1.764 vs 1.660 ms per 1024-frame launch).
// -DHVC_TRAFFIC_ONLY=1 (measurement builds only, never the shipped library): the kernels keep their exact memory
// traffic -- every load, every store, the LDS exchanges -- but skip the arithmetic, so that a run shows the memory
// ceiling of each kernel's OWN access shape.  The outputs are then garbage by construction.
#ifndef HVC_TRAFFIC_ONLY
#define HVC_TRAFFIC_ONLY 0
#endif
#if HVC_TRAFFIC_ONLY
#define HVC_TRAFFIC_ONLY_DECODE(OUT)                                                                    \
    _Pragma("unroll") for (int j = 0; j < 8; j++) {                                                     \
        (OUT)[j][0] = w[4 * j] ^ w[4 * j + 2];                                                          \
        (OUT)[j][1] = w[4 * j + 1] ^ w[4 * j + 3];                                                      \
    }                                                                                                   \
    break;
#else
#define HVC_TRAFFIC_ONLY_DECODE(OUT)
#endif
// PRE: statements run on the loaded dwords w[] before anything reads them (the DC override of DecodeParams::dc_plane).
// -DHVC_TRAFFIC_ONLY=2: additionally, the loads of a wavefront cover whole lines (lane l takes piece l of 1 KB per
// instruction instead of 16 bytes at a stride of 128): what a load shape with a transpose behind it could reach.
#if HVC_TRAFFIC_ONLY == 2
#define HVC_PACKED_SRC(SRC, j) (SRC)[hvc_src_stride * (j)]
#else
#define HVC_PACKED_SRC(SRC, j) (SRC)[j]
#endif
#define HVC_DECODE_BLOCK_PACKED(SRC, QP, OUT, G, PRE)                                                   \
    do {                                                                                                \
        unsigned w[32];                                                                                 \
        _Pragma("unroll") for (int j = 0; j < 8; j++) {                                                 \
            const uint4 t = HVC_PACKED_SRC(SRC, j);                                                     \
            w[4 * j + 0] = t.x;                                                                         \
            w[4 * j + 1] = t.y;                                                                         \
            w[4 * j + 2] = t.z;                                                                         \
            w[4 * j + 3] = t.w;                                                                         \
        }                                                                                               \
        PRE                                                                                             \
        HVC_TRAFFIC_ONLY_DECODE(OUT)                                                                    \
        _Pragma("unroll") for (int d = 0; d < 32; d++) (G).energy = dot2_sat(w[d], w[d], (G).energy);   \
        /* rows in the order that completes one column operand pair per two rows */                    \
        unsigned cA[8], cB[8], cC[8], cZ[8];                                                            \
        {                                                                                               \
            int ra[8], rb[8];                                                                           \
            idct_row_packed<HVC_PAIR_A_LO>(w, (QP) + 4 * HVC_PAIR_A_LO, ra, (G));                       \
            idct_row_packed<HVC_PAIR_A_HI>(w, (QP) + 4 * HVC_PAIR_A_HI, rb, (G));                       \
            _Pragma("unroll") for (int c = 0; c < 8; c++) cA[c] = pack_rows(ra[c], rb[c], (G)); /* (x4, x5) = (r1, r7) */ \
            idct_row_packed<HVC_PAIR_B_LO>(w, (QP) + 4 * HVC_PAIR_B_LO, ra, (G));                       \
            idct_row_packed<HVC_PAIR_B_HI>(w, (QP) + 4 * HVC_PAIR_B_HI, rb, (G));                       \
            _Pragma("unroll") for (int c = 0; c < 8; c++) cB[c] = pack_rows(ra[c], rb[c], (G)); /* (x6, x7) = (r5, r3) */ \
            idct_row_packed<HVC_PAIR_C_LO>(w, (QP) + 4 * HVC_PAIR_C_LO, ra, (G));                       \
            idct_row_packed<HVC_PAIR_C_HI>(w, (QP) + 4 * HVC_PAIR_C_HI, rb, (G));                       \
            _Pragma("unroll") for (int c = 0; c < 8; c++) cC[c] = pack_rows(ra[c], rb[c], (G)); /* (x3, x2) = (r2, r6) */ \
            idct_row_packed<HVC_PAIR_Z_LO>(w, (QP) + 4 * HVC_PAIR_Z_LO, ra, (G));                       \
            idct_row_packed<HVC_PAIR_Z_HI>(w, (QP) + 4 * HVC_PAIR_Z_HI, rb, (G));                       \
            _Pragma("unroll") for (int c = 0; c < 8; c++) cZ[c] = pack_rows(ra[c], rb[c], (G)); /* (b0, b4) = (r0, r4) */ \
        }                                                                                               \
        /* columns two at a time: 16 results -> 8 row halves of the output dwords */                   \
        _Pragma("unroll") for (int c = 0; c < 8; c += 2) {                                              \
            int ca[8], cb[8];                                                                           \
            idct_col_packed(cA[c], cB[c], cC[c], cZ[c], ca, (G));                                       \
            idct_col_packed(cA[c + 1], cB[c + 1], cC[c + 1], cZ[c + 1], cb, (G));                       \
            _Pragma("unroll") for (int j = 0; j < 8; j++) {                                             \
                if ((c & 2) == 0)                                                                       \
                    ashr14_sat_pack2<0>((OUT)[j][c >> 2], ca[j], cb[j]);                                \
                else                                                                                    \
                    ashr14_sat_pack2<1>((OUT)[j][c >> 2], ca[j], cb[j]);                                \
            }                                                                                           \
        }                                                                                               \
    } while (0)

__device__ __forceinline__ bool packed_guard_failed(const PackedGuard &g, int ethr) {
    return (g.energy > ethr) | (g.renergy >= GUARD_RE) | (g.ymax > GUARD_Y) | (g.ymin < -GUARD_Y);
}

// coefficient 0 (the low half of the first dword) replaced by the block's entry of the compact DC array
__device__ __forceinline__ unsigned with_dc(unsigned w0, int16_t dc) { return (w0 & 0xffff0000u) | (unsigned)(unsigned short)dc; }

// DCP: the DC comes from DecodeParams::dc_plane (the instantiation the batch pipeline behind the GPU Huffman reader
// launches); the default instantiation is the kernel as it always was.
template <bool DCP>
__global__ __launch_bounds__(HVC_PACKED_LB) void k_decode_packed(DecodeParams P) {
    BlockRef br;
    const int lane = threadIdx.x;
    unsigned wframe, wtile;
    xcd_work(P.xcd_map, P.xcd_magic, wframe, wtile);
    const bool active = locate(P, (int)wframe, (int)wtile, lane, br);
    const uint4 *src = reinterpret_cast<const uint4 *>(P.coefs + br.coef_idx);
    const unsigned *__restrict__ qp = P.qpair + br.qtab * 32; // wave-uniform, kernarg segment
    PackedGuard g;
    unsigned out[8][2];
    int16_t dcv = 0;
    if (DCP) dcv = P.dc_plane[(size_t)wframe * P.dc_fs + ((br.coef_idx - (size_t)wframe * P.coef_fs) >> 6)];
#if HVC_TRAFFIC_ONLY == 2 // whole-line loads where all 64 blocks of the wavefront exist (they are contiguous then)
    const bool hvc_full = __ballot(active) == ~0ull;
    const int hvc_src_stride = hvc_full ? 64 : 1;
    if (hvc_full) src = src - 8 * (lane & 63) + (lane & 63);
#endif
    HVC_DECODE_BLOCK_PACKED(src, qp, out, g, if (DCP) w[0] = with_dc(w[0], dcv););
#ifdef HVC_K1_PAD_VALU /* probe only (profiles/r05g_k1_valu_probe.txt): N more slow-issue VALU instructions per lane -- how
                          much of K1's time is its instruction issue? */
    {
        int pad = lane;
#pragma unroll
        for (int i = 0; i < HVC_K1_PAD_VALU; i++) asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(pad) : "v"(lane));
        asm volatile("" ::"v"(pad));
    }
#endif

    const bool bad = packed_guard_failed(g, P.ethr_packed[br.qtab]);
    if (active && !bad) {
#pragma unroll
        for (int j = 0; j < 8; j++)
            store_row8(P.pixels + br.pix_idx + (size_t)j * br.stride, out[j][0], out[j][1]);
    }
    const bool flag = active && bad;
    const unsigned long long m = __ballot(flag);
    if (m) {
        const int wl = lane & 63;
        unsigned base = 0;
        if (wl == 0) base = atomicAdd(P.fix_count, (unsigned)__popcll(m));
        base = __shfl(base, 0);
        if (flag) {
            unsigned idx = base + (unsigned)__popcll(m & ((1ull << wl) - 1ull));
            P.fix_list[idx] = (wframe * (unsigned)P.tiles_per_frame + wtile) * HVC_TILE + lane;
        }
    }
}

// ---------------------------------------------------------------------------
// K1 q16: the mapping BASELINE.json's north star describes -- one 8x8 block per QUARTER WAVEFRONT
// (16 lanes), coefficients staged in LDS between the passes, coalesced loads of the zig-zag
// records -- kept as a selectable alternative (hvc_set_decode_kernel(ctx, 3)) so that the choice of
// the block-per-lane kernel rests on a measurement, not an argument.  Same arithmetic and the same
// guard as k_decode_packed (int16 operand pairs + v_dot2_i32_i16), so the same interval proof holds.
//   load     lane l of a group reads zig-zag coefficients 4l .. 4l+3 (8 B; 128 B contiguous per block,
//            16 adjacent blocks = 2 KiB per workgroup iteration), dequantises them (v_pk_mul_lo_u16)
//            and scatters them (inverse zig-zag) into the block's LDS stage as int16
//   row pass lane (r = l >> 1, h = l & 1): the even half owns (b0,b4) (b2,b6), the odd half
//            (b1,b7) (b5,b3): both run the SAME instruction stream -- two rotations by v_dot2 with
//            per-lane constant pairs, one butterfly; only the odd half keeps the 181 stage (select) --
//            then swap their four values with the partner lane (DPP quad_perm) and form sums (even)
//            or differences (odd): row outputs 0,3,1,2 / 7,4,6,5
//   transpose through LDS, column pass the same way, >> 14 / clip / +128 by v_ashr_pk_u8_i32
//   store    pixels collect in a 16 KiB LDS tile (256 blocks); after 16 iterations the workgroup
//            writes it with K1's store shape (8 x 8 B per lane, non-temporal)
constexpr int Q16_ITERS = HVC_TILE / 16;

__device__ __forceinline__ int dot2vv(unsigned pair, unsigned k, int add) {
    int d;
    asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(d) : "v"(pair), "v"(k), "v"(add));
    return d;
}
// value of the partner lane (l ^ 1)
__device__ __forceinline__ int partner(int v) { return __builtin_amdgcn_mov_dpp(v, 0xB1, 0xf, 0xf, true); } // quad_perm:[1,0,3,2]

struct Q16Lane {
    unsigned k00, k01, k10, k11; // constant pairs of the two rotations
    int a0, a1, sh0, sh1;        // addend / right shift of each rotation's outputs
};

// One half-butterfly + exchange.  P0 / P1: the lane's two operand pairs.  Returns the lane's four
// outputs (even half: positions 0,3,1,2; odd half: 7,4,6,5), unshifted.
__device__ __forceinline__ void q16_pass(unsigned P0, unsigned P1, const Q16Lane &K, bool odd, int negm, int (&o)[4],
                                         int &ymax, int &ymin) {
    const int p = dot2vv(P0, K.k00, K.a0) >> K.sh0, q = dot2vv(P0, K.k01, K.a0) >> K.sh0;
    const int s = dot2vv(P1, K.k10, K.a1) >> K.sh1, t = dot2vv(P1, K.k11, K.a1) >> K.sh1;
    const int u1 = p + s, u2 = p - s, u3 = q + t, u4 = q - t;
    // odd half: x2 = (181 (x4 + x5) + 128) >> 8, x4 = (181 (x4 - x5) + 128) >> 8   (dct.ml:41-42, 83-84)
    const int ys = odd ? u2 + u4 : 0, yd = odd ? u2 - u4 : 0;
    ymax = max(max(ymax, ys), yd);
    ymin = min(min(ymin, ys), yd);
    const int w2 = mad24(181, ys, 128) >> 8, w4 = mad24(181, yd, 128) >> 8;
    // even: (x7, x8, x3, x0)   odd: (x1, x6, x2, x4)   -> outputs (0|7, 3|4, 1|6, 2|5)
    const int m0 = u1, m1 = odd ? u3 : u2, m2 = odd ? w2 : u3, m3 = odd ? w4 : u4;
    // even lanes add the partner's values, odd lanes subtract their own from the partner's:
    // theirs + (mine ^ negm) - negm with negm = odd ? -1 : 0
    o[0] = partner(m0) + ((m0 ^ negm) - negm);
    o[1] = partner(m1) + ((m1 ^ negm) - negm);
    o[2] = partner(m2) + ((m2 ^ negm) - negm);
    o[3] = partner(m3) + ((m3 ^ negm) - negm);
}

__global__ __launch_bounds__(HVC_TILE) void k_decode_q16(DecodeParams P) {
    __shared__ short stage[16][64];          // per group: the block's 64 int16 operands of the next pass
    __shared__ uint8_t tile[8][HVC_TILE][8]; // pixel rows of the workgroup's 256 blocks
    __shared__ unsigned short qlds[64];      // the component's quantiser table (zig-zag order)
    __shared__ uint8_t badflag[HVC_TILE];

    const int lane = threadIdx.x, g = lane >> 4, l = lane & 15;
    BlockRef br; // of block `lane` of the tile: used by the store phase
    const bool active = locate(P, blockIdx.y, blockIdx.x, lane, br);
    if (lane < 64) qlds[lane] = (unsigned short)P.qt[br.qtab * 64 + lane]; // kernarg -> LDS once per tile
    __syncthreads();

    const bool odd = l & 1;
    const int negm = odd ? -1 : 0;
    const int r = l >> 1; // row in the row pass, column in the column pass
    Q16Lane KR, KC;
    KR.k10 = odd ? pk(W5, W3) : pk(W2, W6);
    KR.k11 = odd ? pk(W3, -W5) : pk(W6, -W2);
    KR.k00 = odd ? pk(W1, W7) : pk(2048, 2048);
    KR.k01 = odd ? pk(W7, -W1) : pk(2048, -2048);
    KR.a0 = odd ? 0 : 128;
    KR.a1 = 0;
    KR.sh0 = KR.sh1 = 0;
    KC.k10 = KR.k10;
    KC.k11 = KR.k11;
    KC.k00 = odd ? pk(W1, W7) : pk(256, 256);
    KC.k01 = odd ? pk(W7, -W1) : pk(256, -256);
    KC.a0 = odd ? 4 : 8192 + (128 << 14);
    KC.a1 = 4;
    KC.sh0 = odd ? 3 : 0;
    KC.sh1 = 3;
    const unsigned q01 = qlds[4 * l] | ((unsigned)qlds[4 * l + 1] << 16), q23 = qlds[4 * l + 2] | ((unsigned)qlds[4 * l + 3] << 16);

    // operand slot of raster position (row, col) for the pass that runs along `row`:
    // lane 2*row + (col & 1); even lanes hold [c0 c4 | c2 c6], odd lanes [c1 c7 | c5 c3]
    auto slot = [](int row, int col) {
        const int pos = (col & 1) ? ((col == 1) ? 0 : (col == 7) ? 1 : (col == 5) ? 2 : 3)
                                  : ((col == 0) ? 0 : (col == 4) ? 1 : (col == 2) ? 2 : 3);
        return (2 * row + (col & 1)) * 4 + pos;
    };
    int ld_slot[4]; // where this lane's four loaded coefficients (zig-zag 4l + i) go for the row pass
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int p = ZI[4 * l + i];
        ld_slot[i] = slot(p >> 3, p & 7);
    }
    // this lane's four row outputs are columns (0,3,1,2) / (7,4,6,5) of row r: slots for the column pass
    // (the column pass runs along the column, so the roles of row and column swap)
    const int oc[4] = {odd ? 7 : 0, odd ? 4 : 3, odd ? 6 : 1, odd ? 5 : 2};
    int tr_slot[4];
#pragma unroll
    for (int i = 0; i < 4; i++) tr_slot[i] = slot(oc[i], r);

    const size_t plane_coefs = br.plane_coef_idx;
    for (int it = 0; it < Q16_ITERS; it++) {
        const int bi = it * 16 + g;                                  // block inside the tile
        int b = br.tile_b0 + bi;
        const bool exists = b < br.nblk;
        b = exists ? b : br.nblk - 1;
        const uint2 cz = reinterpret_cast<const uint2 *>(P.coefs + plane_coefs + (size_t)b * 64)[l];
        int energy = dot2_sat(cz.y, cz.y, dot2_sat(cz.x, cz.x, 0));
        const unsigned d01 = pk_mul_lo(cz.x, q01), d23 = pk_mul_lo(cz.y, q23);
        short *st = stage[g];
        st[ld_slot[0]] = (short)(d01 & 0xffffu);
        st[ld_slot[1]] = (short)(d01 >> 16);
        st[ld_slot[2]] = (short)(d23 & 0xffffu);
        st[ld_slot[3]] = (short)(d23 >> 16);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        uint2 ops = reinterpret_cast<const uint2 *>(st)[l];
        int ymax = 0, ymin = 0, o[4];
        q16_pass(ops.x, ops.y, KR, odd, negm, o, ymax, ymin);
        // row outputs >> 8, saturating int16 (the energy of the saturated values is the guard)
        const unsigned r01 = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pk_i16(o[0] >> 8, o[1] >> 8));
        const unsigned r23 = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pk_i16(o[2] >> 8, o[3] >> 8));
        int renergy = dot2_sat(r23, r23, dot2_sat(r01, r01, 0));
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier(); // every lane has read its operands before the stage is overwritten
        st[tr_slot[0]] = (short)(r01 & 0xffffu);
        st[tr_slot[1]] = (short)(r01 >> 16);
        st[tr_slot[2]] = (short)(r23 & 0xffffu);
        st[tr_slot[3]] = (short)(r23 >> 16);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        ops = reinterpret_cast<const uint2 *>(st)[l];
        q16_pass(ops.x, ops.y, KC, odd, negm, o, ymax, ymin);
        __builtin_amdgcn_wave_barrier();
        // guard: the two energies are sums over the group's 16 lanes -- quad_perm, quad_perm,
        // row_half_mirror, row_mirror butterflies (DPP; no LDS), saturating at 2^30 (> both thresholds,
        // so a clipped sum still compares the right way; 2^30 + 2^30 fits an unsigned dword).
        constexpr unsigned CAP = 1u << 30;
        unsigned es = min((unsigned)energy, CAP), rs = min((unsigned)renergy, CAP);
#define HVC_Q16_STEP(CTRL)                                                                       \
        es = min(es + (unsigned)__builtin_amdgcn_mov_dpp((int)es, CTRL, 0xf, 0xf, true), CAP); \
        rs = min(rs + (unsigned)__builtin_amdgcn_mov_dpp((int)rs, CTRL, 0xf, 0xf, true), CAP);
        HVC_Q16_STEP(0xB1)  // lane ^ 1
        HVC_Q16_STEP(0x4E)  // lane ^ 2
        HVC_Q16_STEP(0x141) // the other quad of the 8
        HVC_Q16_STEP(0x140) // the other half of the 16
#undef HVC_Q16_STEP
        const bool lbad = (es > (unsigned)P.ethr_packed[br.qtab]) | (rs >= (unsigned)GUARD_RE) | (ymax > GUARD_Y) |
                          (ymin < -GUARD_Y);
        const bool bad = ((__ballot(lbad) >> (lane & 48)) & 0xffffull) != 0; // any lane of the group
        // column outputs: rows (0,3,1,2) / (7,4,6,5) of column r of the block
        unsigned px01 = 0, px23 = 0;
        ashr14_sat_pack2<0>(px01, o[0], o[1]);
        ashr14_sat_pack2<0>(px23, o[2], o[3]);
        tile[oc[0]][bi][r] = (uint8_t)(px01 & 0xffu);
        tile[oc[1]][bi][r] = (uint8_t)((px01 >> 8) & 0xffu);
        tile[oc[2]][bi][r] = (uint8_t)(px23 & 0xffu);
        tile[oc[3]][bi][r] = (uint8_t)((px23 >> 8) & 0xffu);
        if (l == 0) badflag[bi] = (uint8_t)(bad && exists);
    }
    __syncthreads();
    const bool bad = badflag[lane] != 0;
    if (active && !bad) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const uint2 px = *reinterpret_cast<const uint2 *>(&tile[j][lane][0]);
            store_row8(P.pixels + br.pix_idx + (size_t)j * br.stride, px.x, px.y);
        }
    }
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

// K1 wide: the model's own 63-bit arithmetic.  Two uses: the fix-up list of a launch (blocks the packed kernel's guard
// sent here: normally none) and WHOLE calls -- a quantiser entry above 255 (16-bit DQT) or hvc_set_decode_kernel(ctx, 2)
// sends every block through it -- so it is built on K1's skeleton: one block per lane, the block's 128 bytes as 8 x 16 B
// loads, every intermediate in registers (the 64 int64 values of the transposed intermediate are 128 VGPRs; every index
// below is a compile-time constant after unrolling, nothing goes to scratch), 8 x 8-byte row stores.
//
// One block: w = its 32 coefficient dwords (zig-zag order as loaded), q = its table (zig-zag order); has_dc: the DC is
// `dc` (the model's 63-bit number: DecodeParams::dc_plane or the side list of DCs beyond int16) instead of coefficient 0.
// decoder.ml:142-149 (dequantise + inverse zig-zag), dct.ml:11-107 (rows, then columns), decoder.ml:213-224 (clip, + 128).
template <bool WRAP63>
__device__ __forceinline__ void decode_block_wide(const unsigned (&w)[32], const int *__restrict__ q, bool has_dc, int64_t dc,
                                                  unsigned (&out)[8][2]) {
    int64_t v[64];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        uint64_t in[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int zz = ZF[8 * r + i]; // raster 8r + i sits at zig-zag position zz: dword zz / 2, half zz & 1
            const int c = (zz & 1) ? (int)w[zz >> 1] >> 16 : (int)(short)(w[zz >> 1] & 0xffffu);
            in[i] = (uint64_t)((int64_t)c * (int64_t)q[zz]);
        }
        if (r == 0 && has_dc) in[0] = (uint64_t)mul63(dc, (int64_t)q[0]);
        int64_t o[8];
        idct8_wide<false, WRAP63>(in, o);
#pragma unroll
        for (int i = 0; i < 8; i++) v[8 * r + i] = o[i];
    }
#pragma unroll
    for (int j = 0; j < 8; j++) out[j][0] = out[j][1] = 0;
#pragma unroll
    for (int c = 0; c < 8; c++) {
        uint64_t in[8];
#pragma unroll
        for (int j = 0; j < 8; j++) in[j] = (uint64_t)v[8 * j + c];
        int64_t o[8];
        idct8_wide<true, WRAP63>(in, o);
#pragma unroll
        for (int j = 0; j < 8; j++) { // clip to -128 .. 127, + 128 (decoder.ml:213-224) = the shifted value held to 0 .. 255
            const uint64_t y = (uint64_t)o[j] + 128u;
            const unsigned px = y <= 255u ? (unsigned)y : ((int64_t)y < 0 ? 0u : 255u);
            out[j][c >> 2] |= px << (8 * (c & 3));
        }
    }
}

__device__ __forceinline__ void load_block_dwords(const int16_t *cf, unsigned (&w)[32]) {
    const uint4 *src = reinterpret_cast<const uint4 *>(cf);
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const uint4 t = src[j];
        w[4 * j + 0] = t.x;
        w[4 * j + 1] = t.y;
        w[4 * j + 2] = t.z;
        w[4 * j + 3] = t.w;
    }
}

// The fix-up form: the listed blocks, one per lane, in a grid-stride loop over the (normally empty) list.
// dc_list (optional, parallel to list): the listed block's true absolute DC where it does not fit the int16 record
// (hvc_hdec.h WideDc: the model's 63-bit dc of decoder.ml:143)
// Launched with a fixed grid (the list's length is known on the device only): HVC_FIXUP_WGS workgroups of HVC_FIXUP_LANES
// lanes.  Normally the list is empty and every lane leaves after one load; when an adversarial batch sends EVERY block here
// (tools/bench_configs.py --config 14) the grid decides the cost: 256 x 64 lanes -- one wave per CU -- read 4.5 % of the HBM
// peak on such a batch, 512 x 256 -- the two waves per SIMD the kernel's registers allow -- 9.1 %, with nothing to see in the
// headline (profiles/r06f_fixup_grid.txt).  What is left there is the packed kernel's own worst case (one atomic per
// wavefront on one counter, the list's stores), not this kernel's arithmetic (k_decode_wide_all: 31 - 34 %).
#ifndef HVC_FIXUP_WGS
#define HVC_FIXUP_WGS 512
#endif
#ifndef HVC_FIXUP_LANES
#define HVC_FIXUP_LANES 256
#endif
__global__ __launch_bounds__(HVC_FIXUP_LANES) void k_decode_wide(DecodeParams P, const unsigned *count, const unsigned *list,
                                                                 const long long *dc_list) {
    const unsigned long long n = (unsigned long long)*count;
    // The two fix-up counters alternate between calls: this launch reads the
    // current one and clears the other for the next call (no memset node).
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (P.fix_count_next) *P.fix_count_next = 0;
        // hvc_last_wide_blocks: the call's total (1 = this launch starts it, 0 = adds in stream order, 2 = adds beside
        // another stream's launch)
        if (P.wide_total) {
            if (P.wide_first == 2) atomicAdd(P.wide_total, n);
            else *P.wide_total = P.wide_first ? n : *P.wide_total + n;
        }
    }
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (unsigned long long)gridDim.x * blockDim.x) {
        const unsigned long long id = list[i];
        const int lane = (int)(id % HVC_TILE);
        const unsigned long long t = id / HVC_TILE;
        const int tile = (int)(t % (unsigned)P.tiles_per_frame);
        const int frame = (int)(t / (unsigned)P.tiles_per_frame);
        if (frame >= P.n_frames) continue; // an id from another geometry must never turn into an address
        BlockRef br;
        if (!locate(P, frame, tile, lane, br)) continue;
        unsigned w[32], out[8][2];
        load_block_dwords(P.coefs + br.coef_idx, w);
        int64_t dc = 0;
        if (P.dc_plane) // the DC lives in the compact array (DecodeParams::dc_plane)
            dc = (int64_t)P.dc_plane[(size_t)frame * P.dc_fs + ((br.coef_idx - (size_t)frame * P.coef_fs) >> 6)];
        if (dc_list) dc = (int64_t)dc_list[i];
        decode_block_wide<true>(w, P.qt + br.qtab * 64, P.dc_plane != nullptr || dc_list != nullptr, dc, out);
#pragma unroll
        for (int j = 0; j < 8; j++) store_row8(P.pixels + br.pix_idx + (size_t)j * br.stride, out[j][0], out[j][1]);
    }
}

// Every block of a launch: K1's work decomposition (grid = tiles x frames, xcd_work, one block per lane).
__global__ __launch_bounds__(HVC_TILE) void k_decode_wide_all(DecodeParams P) {
    BlockRef br;
    const int lane = threadIdx.x;
    unsigned wframe, wtile;
    xcd_work(P.xcd_map, P.xcd_magic, wframe, wtile);
    const bool active = locate(P, (int)wframe, (int)wtile, lane, br);
    unsigned w[32], out[8][2];
    load_block_dwords(P.coefs + br.coef_idx, w);
    int64_t dc = 0;
    if (P.dc_plane) dc = (int64_t)P.dc_plane[(size_t)wframe * P.dc_fs + ((br.coef_idx - (size_t)wframe * P.coef_fs) >> 6)];
    decode_block_wide<false>(w, P.qt + br.qtab * 64, P.dc_plane != nullptr, dc, out); // (int16 x 16-bit inputs: asr63's comment)
    if (active) {
#pragma unroll
        for (int j = 0; j < 8; j++) store_row8(P.pixels + br.pix_idx + (size_t)j * br.stride, out[j][0], out[j][1]);
    }
}

// ---------------------------------------------------------------------------
// K1 + crop + K2 fused: 4:2:0 coefficient records -> tight 4:4:4 frames
// (Decoder.decode -> get_yuv_frame, decoder.ml:403-420 -> Planar_444.convert_from_420,
// tools/src/planar_444.ml:82-131).  The quarter-resolution chroma planes never reach HBM.
//
// Luma: the linear 256-block tiles of k_decode_packed over the block rows / columns that intersect
// the crop, rows below the crop not stored.  Chroma: a workgroup owns 64 x 4 blocks (one block row
// per wave).  supersample_hv2 needs, for every source sample a, its right / lower / lower-right
// neighbours b, c, d: inside a block they are in the lane's own registers; across blocks each lane
// publishes its block's first row and first column through LDS (16 B per lane, one barrier).
//   avg2(a,b) = (a+b+1)>>1           = v_lerp_u8(a, b, 0x01010101) on four samples at once
//   avg4(a,b,c,d) = (a+b+c+d+2)>>2   = v_lerp_u8(avg2(a,b), (c+d)>>1, r),
//                                       r = ~(a^b) | (c^d)   (bit 0 of each byte is used)
// (tests/test_guard_bounds.py::test_avg4_by_lerp_identity proves the identity over all sums.)
// Across workgroups: horizontally, consecutive tiles overlap by one block column (lane 63's block is
// decoded twice, 1.6 % more chroma work, so no tile ever needs a block it did not decode);
// vertically, the output row below every 32nd source row is left to k_reinterp_444, which
// recomputes interpolated samples from the source samples already sitting at the even output
// coordinates (coalesced: two rows read, one written).  The same pass rebuilds the surroundings of
// blocks that failed the int32 guard (their source samples are written by k_decode_wide_444 first).
struct Ref444 {
    int p, bx, by;
    bool active; // the lane's block exists
    bool store;  // ... and this lane owns its output (false for a chroma tile's overlap column)
};

// wgs = lanes per workgroup (256 * P.nw), tw = chroma tile width in blocks (64 * P.nw)
__device__ __forceinline__ Ref444 locate444(const Decode444Params &P, int tile, int lane, int wgs, int tw) {
    Ref444 r;
    if (tile < P.y_tiles) {
        const Plane444K &K = P.pl[0];
        const int n = K.cbw * K.cbh;
        int b = tile * wgs + lane;
        r.active = b < n;
        b = r.active ? b : n - 1;
        const unsigned by = K.cbw == 1 ? (unsigned)b : __umulhi((unsigned)b, P.y_magic);
        r.p = 0;
        r.by = (int)by;
        r.bx = b - (int)by * K.cbw;
        r.store = r.active;
    } else {
        int t = tile - P.y_tiles;
        const int per = P.c_tiles_x * P.c_tiles_y;
        r.p = 1;
        if (t >= per) {
            r.p = 2;
            t -= per;
        }
        const unsigned ty = P.c_tiles_x == 1 ? (unsigned)t : __umulhi((unsigned)t, P.c_magic);
        const unsigned tx = (unsigned)t - ty * (unsigned)P.c_tiles_x;
        const Plane444K &K = P.pl[r.p];
        // lanes run along the tile's block row first (tw = a power of two): lane = y_in_tile * tw + x_in_tile.
        // Consecutive tiles share one block column: the last lane of a tile row only feeds its left neighbour's
        // right-neighbour samples, its own output belongs to lane 0 of the next tile (unless it is the last column)
        const int lx = lane & (tw - 1), ly = lane / tw;
        const int bx = (int)tx * (tw - 1) + lx, by = (int)ty * HVC_444_TILE_BH + ly;
        r.active = bx < K.cbw && by < K.cbh;
        r.store = r.active && (lx < tw - 1 || bx == K.cbw - 1);
        r.bx = min(bx, K.cbw - 1);
        r.by = min(by, K.cbh - 1);
    }
    return r;
}

template <bool ALIGNED>
__device__ __forceinline__ void store16(uint8_t *row, int x0, int w, unsigned d0, unsigned d1, unsigned d2,
                                        unsigned d3) {
    if (ALIGNED) {
        typedef unsigned u4s __attribute__((ext_vector_type(4)));
        const u4s t = {d0, d1, d2, d3};
        __builtin_nontemporal_store(t, reinterpret_cast<u4s *>(row + x0));
    } else {
        const unsigned d[4] = {d0, d1, d2, d3};
        for (int i = 0; i < 16; i++)
            if (x0 + i < w) row[x0 + i] = (uint8_t)(d[i >> 2] >> (8 * (i & 3)));
    }
}

// One source row of a chroma block: the samples a, the rounded-up and rounded-down averages with the
// right neighbour, and a ^ b.
struct RowQ {
    unsigned a0, a1, hc0, hc1, hf0, hf1, x0, x1;
};
// NEXT = the dword holding the sample right of a's last one, in byte K; m0/m1 = bytes at or beyond
// the plane's last column (there b = a: supersample_hv2's "w - 1" column, planar_444.ml:97-102)
template <int K>
__device__ __forceinline__ RowQ rowq(unsigned a0, unsigned a1, unsigned next, unsigned m0, unsigned m1) {
    RowQ q;
    q.a0 = a0;
    q.a1 = a1;
    unsigned b0 = __builtin_amdgcn_alignbyte(a1, a0, 1);
    unsigned b1 = __builtin_amdgcn_perm(next, a1, ((unsigned)(4 + K) << 24) | 0x030201u);
    b0 = (m0 & a0) | (~m0 & b0);
    b1 = (m1 & a1) | (~m1 & b1);
    q.hc0 = __builtin_amdgcn_lerp(a0, b0, 0x01010101u);
    q.hc1 = __builtin_amdgcn_lerp(a1, b1, 0x01010101u);
    q.hf0 = __builtin_amdgcn_lerp(a0, b0, 0u);
    q.hf1 = __builtin_amdgcn_lerp(a1, b1, 0u);
    q.x0 = a0 ^ b0;
    q.x1 = a1 ^ b1;
    return q;
}

// output rows 2y and 2y + 1 of a chroma block's source row `cur` (next = source row y + 1)
template <bool ALIGNED>
__device__ __forceinline__ void emit_rows444(uint8_t *row_even, size_t W, int x0, const RowQ &cur, const RowQ &nxt,
                                             bool last_row) {
    constexpr unsigned LO = 0x05010400u, HI = 0x07030602u; // interleave (a, h): a.b0 h.b0 a.b1 h.b1 / a.b2 ...
    store16<ALIGNED>(row_even, x0, (int)W, __builtin_amdgcn_perm(cur.hc0, cur.a0, LO),
                     __builtin_amdgcn_perm(cur.hc0, cur.a0, HI), __builtin_amdgcn_perm(cur.hc1, cur.a1, LO),
                     __builtin_amdgcn_perm(cur.hc1, cur.a1, HI));
    unsigned v0 = __builtin_amdgcn_lerp(cur.a0, nxt.a0, 0x01010101u);
    unsigned v1 = __builtin_amdgcn_lerp(cur.a1, nxt.a1, 0x01010101u);
    unsigned q0 = __builtin_amdgcn_lerp(cur.hc0, nxt.hf0, ~cur.x0 | nxt.x0);
    unsigned q1 = __builtin_amdgcn_lerp(cur.hc1, nxt.hf1, ~cur.x1 | nxt.x1);
    if (last_row) { // row2 = min (h - 1) (row + 1) = row (planar_444.ml:86): avg2 a a = a, avg4 a b a b = avg2 a b
        v0 = cur.a0;
        v1 = cur.a1;
        q0 = cur.hc0;
        q1 = cur.hc1;
    }
    store16<ALIGNED>(row_even + W, x0, (int)W, __builtin_amdgcn_perm(q0, v0, LO), __builtin_amdgcn_perm(q0, v0, HI),
                     __builtin_amdgcn_perm(q1, v1, LO), __builtin_amdgcn_perm(q1, v1, HI));
}

#ifdef HVC_444_WAVES /* experiments: -DHVC_444_WAVES=4 */
#define HVC_444_ATTR __attribute__((amdgpu_waves_per_eu(HVC_444_WAVES, HVC_444_WAVES)))
#else
#define HVC_444_ATTR
#endif
// NW = P.nw: 256 * NW lanes per workgroup (16 waves per CU asked for in every form: NW = 1 -> 4 workgroups, ...)
template <bool ALIGNED, bool DCP, int NW>
__global__ __launch_bounds__(HVC_TILE * NW, 4 / NW) HVC_444_ATTR void k_decode_444(Decode444Params P) {
    constexpr int WGS = HVC_TILE * NW, TW = HVC_444_TILE_BW * NW;
    __shared__ uint4 edge[WGS]; // per lane: first row (x, y) and first column (z, w) of its chroma block
    const int lane = threadIdx.x;
    unsigned wframe, wtile;
    xcd_work(P.xcd_map, P.xcd_magic, wframe, wtile);
    const int tile = (int)wtile + P.tile0;             // (tile0 = y_tiles when the luma planes went through k_decode_packed)
    const bool chroma = tile >= P.y_tiles;             // workgroup-uniform
    if (P.skip && P.skip == (chroma ? 2 : 1)) return;  // (measurements: one half of the kernel alone)
    const Ref444 r = locate444(P, tile, lane, WGS, TW);
    const Plane444K &K = P.pl[r.p];
    const size_t in_frame = K.coef_off + ((size_t)r.by * K.bw + r.bx) * 64; // the block's place in the frame record
    const uint4 *src = reinterpret_cast<const uint4 *>(P.coefs + (size_t)wframe * P.coef_fs + in_frame);
    const unsigned *__restrict__ qp = P.qpair + K.qtab * 32;
    PackedGuard g;
    unsigned out[8][2];
    int16_t dcv = 0;
    if (DCP) dcv = P.dc_plane[(size_t)wframe * P.dc_fs + (in_frame >> 6)];
#if HVC_TRAFFIC_ONLY == 2
    const int hvc_src_stride = 1; // (the fused kernel keeps its load shape)
#endif
    HVC_DECODE_BLOCK_PACKED(src, qp, out, g, if (DCP) w[0] = with_dc(w[0], dcv););
    const bool bad = packed_guard_failed(g, P.ethr_packed[K.qtab]);

    const size_t W = (size_t)P.width;
    uint8_t *plane = P.out + (size_t)wframe * P.out_fs + K.out_off;
    const int lasty = K.ah - 1 - r.by * 8; // last block row that is inside the crop (>= 0 for active lanes)
    if (!chroma) {
        if (r.active && !bad) {
            uint8_t *p = plane + (size_t)r.by * 8 * W + (size_t)r.bx * 8;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                if (j <= lasty) {
                    if (ALIGNED) {
                        store_row8(p + (size_t)j * W, out[j][0], out[j][1]);
                    } else {
                        for (int i = 0; i < 8; i++)
                            if (r.bx * 8 + i < K.aw) p[(size_t)j * W + i] = (uint8_t)(out[j][i >> 2] >> (8 * (i & 3)));
                    }
                }
            }
        }
    } else {
        // first column of the block, rows 0-3 and 4-7
        constexpr unsigned B0 = 0x0c0c0400u, JOIN = 0x05040100u;
        const unsigned c0 = __builtin_amdgcn_perm(__builtin_amdgcn_perm(out[3][0], out[2][0], B0),
                                                  __builtin_amdgcn_perm(out[1][0], out[0][0], B0), JOIN);
        const unsigned c1 = __builtin_amdgcn_perm(__builtin_amdgcn_perm(out[7][0], out[6][0], B0),
                                                  __builtin_amdgcn_perm(out[5][0], out[4][0], B0), JOIN);
        edge[lane] = make_uint4(out[0][0], out[0][1], c0, c1);
        __syncthreads();
        // right / lower / lower-right neighbours; past the tile the values are placeholders (seam pass)
        const uint4 rt = edge[min(lane + 1, WGS - 1)];
        const uint4 dn = edge[min(lane + TW, WGS - 1)];
        const unsigned dg = edge[min(lane + TW + 1, WGS - 1)].x;
        const int lastx = K.aw - 1 - r.bx * 8; // samples at or beyond it take b = a
        const unsigned long long mm = lastx >= 8 ? 0ull : (~0ull << (8 * max(lastx, 0)));
        const unsigned m0 = (unsigned)mm, m1 = (unsigned)(mm >> 32);
        if (r.store && !bad) {
            uint8_t *p = plane + (size_t)r.by * 16 * W;
            const int x0 = r.bx * 16;
            const RowQ q0 = rowq<0>(out[0][0], out[0][1], rt.z, m0, m1);
            const RowQ q1 = rowq<1>(out[1][0], out[1][1], rt.z, m0, m1);
            const RowQ q2 = rowq<2>(out[2][0], out[2][1], rt.z, m0, m1);
            const RowQ q3 = rowq<3>(out[3][0], out[3][1], rt.z, m0, m1);
            const RowQ q4 = rowq<0>(out[4][0], out[4][1], rt.w, m0, m1);
            const RowQ q5 = rowq<1>(out[5][0], out[5][1], rt.w, m0, m1);
            const RowQ q6 = rowq<2>(out[6][0], out[6][1], rt.w, m0, m1);
            const RowQ q7 = rowq<3>(out[7][0], out[7][1], rt.w, m0, m1);
            const RowQ q8 = rowq<0>(dn.x, dn.y, dg, m0, m1);
            if (0 <= lasty) emit_rows444<ALIGNED>(p + 0 * W, W, x0, q0, q1, lasty == 0);
            if (1 <= lasty) emit_rows444<ALIGNED>(p + 2 * W, W, x0, q1, q2, lasty == 1);
            if (2 <= lasty) emit_rows444<ALIGNED>(p + 4 * W, W, x0, q2, q3, lasty == 2);
            if (3 <= lasty) emit_rows444<ALIGNED>(p + 6 * W, W, x0, q3, q4, lasty == 3);
            if (4 <= lasty) emit_rows444<ALIGNED>(p + 8 * W, W, x0, q4, q5, lasty == 4);
            if (5 <= lasty) emit_rows444<ALIGNED>(p + 10 * W, W, x0, q5, q6, lasty == 5);
            if (6 <= lasty) emit_rows444<ALIGNED>(p + 12 * W, W, x0, q6, q7, lasty == 6);
            if (7 <= lasty) emit_rows444<ALIGNED>(p + 14 * W, W, x0, q7, q8, lasty == 7);
        }
    }
    const bool flag = r.store && bad;
    const unsigned long long m = __ballot(flag);
    if (m) {
        const int wl = lane & 63;
        unsigned base = 0;
        if (wl == 0) base = atomicAdd(P.fix_count, (unsigned)__popcll(m));
        base = __shfl(base, 0);
        if (flag) {
            unsigned idx = base + (unsigned)__popcll(m & ((1ull << wl) - 1ull));
            P.fix_list[idx] = (wframe * (unsigned)P.tiles_per_frame + (unsigned)tile) * WGS + lane;
        }
    }
}

// int64 model arithmetic for listed blocks (list == nullptr: every block).  Luma blocks are written
// as they are; a chroma block's 64 samples go to the EVEN output coordinates, where supersample_hv2
// puts the source sample; k_reinterp_444 then rebuilds the interpolated ones around it.
__global__ __launch_bounds__(64) void k_decode_wide_444(Decode444Params P, const unsigned *count, const unsigned *list,
                                                        unsigned long long total, const long long *dc_list) {
    const unsigned long long n = list ? (unsigned long long)*count : total;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (P.fix_count_next) *P.fix_count_next = 0;
        if (P.wide_total && list) {
            if (P.wide_first == 2) atomicAdd(P.wide_total, n);
            else *P.wide_total = P.wide_first ? n : *P.wide_total + n;
        }
    }
    for (unsigned long long i = (unsigned long long)blockIdx.x * 64 + threadIdx.x; i < n;
         i += (unsigned long long)gridDim.x * 64) {
        const unsigned long long id = list ? list[i] : i;
        const unsigned wgs = (unsigned)(HVC_TILE * P.nw);
        const int lane = (int)(id % wgs);
        const unsigned long long t = id / wgs;
        const int tile = (int)(t % (unsigned)P.tiles_per_frame);
        const size_t frame = (size_t)(t / (unsigned)P.tiles_per_frame);
        if (frame >= (size_t)P.n_frames) continue; // an id from another geometry must never turn into an address
        const Ref444 r = locate444(P, tile, lane, (int)wgs, HVC_444_TILE_BW * P.nw);
        if (!r.store) continue;
        const Plane444K &K = P.pl[r.p];
        const size_t in_frame = K.coef_off + ((size_t)r.by * K.bw + r.bx) * 64;
        unsigned w[32], out[8][2];
        load_block_dwords(P.coefs + frame * P.coef_fs + in_frame, w);
        int64_t dc = 0;
        if (P.dc_plane) dc = (int64_t)P.dc_plane[frame * P.dc_fs + (in_frame >> 6)];
        if (dc_list) dc = (int64_t)dc_list[i];
        decode_block_wide<true>(w, P.qt + K.qtab * 64, P.dc_plane != nullptr || dc_list != nullptr, dc, out);
        uint8_t *plane = P.out + frame * P.out_fs + K.out_off;
        const int step = r.p == 0 ? 1 : 2;
#pragma unroll
        for (int j = 0; j < 8; j++)
#pragma unroll
            for (int i2 = 0; i2 < 8; i2++) {
                const int x = r.bx * 8 + i2, y = r.by * 8 + j;
                if (x >= K.aw || y >= K.ah) continue;
                plane[(size_t)(y * step) * (size_t)P.width + (size_t)(x * step)] = (uint8_t)(out[j][i2 >> 2] >> (8 * (i2 & 3)));
            }
    }
}

// tools/src/planar_444.ml avg2 / avg4
__device__ __forceinline__ unsigned avg2u(unsigned a, unsigned b) { return (a + b + 1) >> 1; }          // :4-8
__device__ __forceinline__ unsigned avg4u(unsigned a, unsigned b, unsigned c, unsigned d) { return (a + b + c + d + 2) >> 2; } // :10-16

// supersample_hv2 (planar_444.ml:82-103) of ONE source sample (c, r) of an output plane whose even
// coordinates already hold the source samples: writes (2c+1, 2r), (2c, 2r+1), (2c+1, 2r+1).
__device__ __forceinline__ void reinterp_sample(uint8_t *plane, size_t W, int aw, int ah, int c, int r) {
    const int c2 = min(c + 1, aw - 1), r2 = min(r + 1, ah - 1);
    const unsigned a = plane[(size_t)(2 * r) * W + 2 * c], b = plane[(size_t)(2 * r) * W + 2 * c2];
    const unsigned cc = plane[(size_t)(2 * r2) * W + 2 * c], d = plane[(size_t)(2 * r2) * W + 2 * c2];
    plane[(size_t)(2 * r) * W + 2 * c + 1] = (uint8_t)avg2u(a, b);
    plane[(size_t)(2 * r + 1) * W + 2 * c] = (uint8_t)avg2u(a, cc);
    plane[(size_t)(2 * r + 1) * W + 2 * c + 1] = (uint8_t)avg4u(a, b, cc, d);
}

// Output row 2r + 1 of a horizontal seam (source row r, r + 1 < ah), 8 source samples = 16 output
// bytes per thread with 16-byte accesses: only that row reads the tile below.
__device__ __forceinline__ void reinterp_hseam16(uint8_t *plane, size_t W, int aw, int r, int t) {
    const uint8_t *re = plane + (size_t)(2 * r) * W + 16 * (size_t)t;
    const uint8_t *rn = re + 2 * W;
    const uint4 A = *reinterpret_cast<const uint4 *>(re), C = *reinterpret_cast<const uint4 *>(rn);
    const bool more = 8 * t + 8 < aw;
    const unsigned a[4] = {A.x, A.y, A.z, A.w}, c[4] = {C.x, C.y, C.z, C.w};
    const unsigned an = more ? re[16] : (A.w >> 16) & 0xffu, cn = more ? rn[16] : (C.w >> 16) & 0xffu;
    unsigned o[4];
#pragma unroll
    for (int k = 0; k < 4; k++) { // dword k holds source samples 2k (byte 0) and 2k + 1 (byte 2)
        const unsigned a0 = a[k] & 0xffu, a1 = (a[k] >> 16) & 0xffu, c0 = c[k] & 0xffu, c1 = (c[k] >> 16) & 0xffu;
        const unsigned a2 = k < 3 ? a[k + 1] & 0xffu : an, c2 = k < 3 ? c[k + 1] & 0xffu : cn;
        o[k] = avg2u(a0, c0) | (avg4u(a0, a1, c0, c1) << 8) | (avg2u(a1, c1) << 16) | (avg4u(a1, a2, c1, c2) << 24);
    }
    *reinterpret_cast<uint4 *>(plane + (size_t)(2 * r + 1) * W + 16 * (size_t)t) = make_uint4(o[0], o[1], o[2], o[3]);
}

// Pass 3 of the fused path.  all = 1: every source sample of both chroma planes (after a wide-only
// decode).  Otherwise: the tile seams (source rows 32k + 31) and the 9 x 9
// source samples around every listed chroma block (its own 8 x 8 plus the column / row before it,
// whose interpolated samples read this block).  VEC: rows are 16-byte aligned (width % 16 == 0).
template <bool VEC>
__global__ __launch_bounds__(256) void k_reinterp_444(Decode444Params P, const unsigned *count, const unsigned *list,
                                                      int all) {
    const Plane444K &K1 = P.pl[1];
    const int aw = K1.aw, ah = K1.ah;
    const size_t W = (size_t)P.width;
    const size_t frame = blockIdx.y;
    const long long tid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (all) {
        const long long per = (long long)aw * ah;
        if (tid < 2 * per) {
            const int p = tid >= per ? 2 : 1;
            const long long s = tid - (p - 1) * per;
            reinterp_sample(P.out + frame * P.out_fs + P.pl[p].out_off, W, aw, ah, (int)(s % aw), (int)(s / aw));
        }
        return;
    }
    const int nhs = P.c_tiles_y - 1;
    const int hw = VEC ? aw / 8 : aw; // work items per horizontal seam
    const long long per = (long long)nhs * hw;
    if (tid < 2 * per) {
        const int p = tid >= per ? 2 : 1;
        const long long s = tid - (p - 1) * per;
        uint8_t *plane = P.out + frame * P.out_fs + P.pl[p].out_off;
        const int r = (int)(s / hw) * (8 * HVC_444_TILE_BH) + 8 * HVC_444_TILE_BH - 1;
        if (r + 1 < ah) { // the crop's last row was finished by k_decode_444 itself
            if (VEC)
                reinterp_hseam16(plane, W, aw, r, (int)(s % hw));
            else
                reinterp_sample(plane, W, aw, ah, (int)(s % hw), r);
        }
    }
    // guard failures (rare: never for encoder-produced data)
    const unsigned long long n = (unsigned long long)*count * 81ull;
    const unsigned long long nthreads = (unsigned long long)gridDim.x * gridDim.y * 256ull;
    for (unsigned long long i = (unsigned long long)blockIdx.y * gridDim.x * 256ull + (unsigned long long)tid; i < n;
         i += nthreads) {
        const unsigned long long id = list[i / 81];
        const int k = (int)(i % 81);
        const unsigned wgs = (unsigned)(HVC_TILE * P.nw);
        const int lane = (int)(id % wgs);
        const unsigned long long t = id / wgs;
        const int tile = (int)(t % (unsigned)P.tiles_per_frame);
        const size_t f = (size_t)(t / (unsigned)P.tiles_per_frame);
        if (f >= (size_t)P.n_frames) continue;
        const Ref444 rr = locate444(P, tile, lane, (int)wgs, HVC_444_TILE_BW * P.nw);
        if (!rr.store || rr.p == 0) continue;
        const int c = rr.bx * 8 - 1 + k % 9, r = rr.by * 8 - 1 + k / 9;
        if (c < 0 || r < 0 || c >= aw || r >= ah) continue;
        reinterp_sample(P.out + f * P.out_fs + P.pl[rr.p].out_off, W, aw, ah, c, r);
    }
}

// ---------------------------------------------------------------------------
// K3: level shift + Dct.Chen.forward_8x8 + quantise + zig-zag
//   jpeg/model/src/encoder.ml:81-108, jpeg/model/src/dct.ml:109-196.
// Inputs are 8-bit pixels, so every intermediate is bounded a priori (|p-128| <=
// 128, first pass <= 2^11, second pass <= 2^14; tests/test_guard_bounds.py): int32
// with 24-bit multiplies is exact, no guard and no wide kernel.

// dct.ml:109-112
// The multiplies are issued explicitly: left to hipcc 7.2, constants folded through the 24-bit
// intrinsics come back as explicit 24-bit sign extensions (shift pairs / v_bfe_i32, +150 VALU ops per
// block), and plain 32-bit forms become v_mad_u64_u32.  Constant in an SGPR, operands < 2^15.
__device__ __forceinline__ int vmul24(int k, int x) {
    int d;
    asm("v_mul_i32_i24 %0, %1, %2" : "=v"(d) : "s"(k), "v"(x));
    return d;
}
__device__ __forceinline__ int vmad24(int k, int x, int acc) {
    int d;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(d) : "s"(k), "v"(x), "v"(acc));
    return d;
}
// (Round 4 tried the rotations as one v_dot2_i32_i16 per output on int16 operand pairs packed once per pair: 80 VALU
// instructions per block fewer (1062 -> 982) and 72 VGPRs -- and 0.3 - 0.7 points SLOWER than this form on the same box,
// three alternations: profiles/r04q_k3_dot2_ab.txt.  K3 does not wait for its VALU.)
// What held K3 under its own traffic-only build was the time its instructions take to issue, not its LDS transpose: 827.6 M
// wave-instructions per 256-frame launch -- 97 % of the 1024 SIMDs' cycles at the 4 cycles SQ_ACTIVE_INST_VALU books each at, about
// 70 % by the measured issue times (profiles/r01_valu_ubench.txt) -- with SQ_WAIT_INST_LDS at 0.04 % of the wave cycles and no bank
// conflicts (profiles/r05c_k3_pmc.txt): little slack for 5 waves per SIMD to hide a memory-bound kernel's latencies behind.  It
// sat 1 - 2 points under that traffic-only build.  HVC_ENCODE_MULHI=1 (shipped) takes 80 of the 1062 instructions per block out -- c4 as one
// v_mul_hi_i32_i24 on operands pre-shifted by the add or mad that makes them, the quantiser's products two at a time
// (v_pk_mul_f32) -- and reads 75.3 - 76.7 % where the 0 form reads 73.5 - 74.1 %, its traffic-only build 74.9 - 77.4 %
// (same box, three alternations: profiles/r05c_k3_ab.txt).
#ifndef HVC_ENCODE_MULHI
#define HVC_ENCODE_MULHI 1
#endif
// HVC_ENCODE_QMAGIC=1 (shipped): the quantiser as one v_fma_f32 per coefficient and one v_perm_b32 per pair (below): another
// 31 instructions out, the slow-issue conversions among them (v_cvt_rpi_i32_f32, v_cvt_pk_i16_i32: 1.7 x the issue time of an
// add or fma, profiles/r01_valu_ubench.txt).  Same box, four alternations (profiles/r05d_k3_variants.txt): 73.9 % without
// either, 75.4 % with MULHI, 76.2 % with both (2 = the fma as v_pk_fma_f32: 76.0 %), traffic-only build 74.8 %.
#ifndef HVC_ENCODE_QMAGIC
#define HVC_ENCODE_QMAGIC 1
#endif
// c4 without its shift: (362 x) >> 9 = the high dword of (x << 9) * (362 << 14) as a 24 x 24 -> 48 bit product
// (v_mul_hi_i32_i24: floor, like asr) -- exact while |x << 9| < 2^23, i.e. |x| < 2^14; the sums c4 sees are <= 1024 in the
// column pass and <= 8 * 724 = 5792 in the row pass (tests/test_guard_bounds.py).  The << 9 rides in the producing add
// (v_add_lshl_u32) or mad, so a c4 is 2 instructions instead of 3.
__device__ __forceinline__ int vmulhi24(int k, int x) {
    int d;
    asm("v_mul_hi_i32_i24 %0, %1, %2" : "=v"(d) : "s"(k), "v"(x));
    return d;
}
__device__ __forceinline__ int c4(int f, int g) { return vmul24(362, f + g) >> 9; }
__device__ __forceinline__ int c4m(int f, int g) { return vmul24(362, f - g) >> 9; }
__device__ __forceinline__ int c62(int f, int g) { return vmad24(473, g, vmul24(196, f)) >> 9; }
__device__ __forceinline__ int c71(int f, int g) { return vmad24(502, g, vmul24(100, f)) >> 9; }
__device__ __forceinline__ int c35(int f, int g) { return vmad24(284, g, vmul24(426, f)) >> 9; }
// c62 b3 (-b2), c35 a2 (-a1), c71 a3 (-a0): the negation rides in the constant
__device__ __forceinline__ int c62n(int f, int g) { return vmad24(-473, g, vmul24(196, f)) >> 9; }
__device__ __forceinline__ int c71n(int f, int g) { return vmad24(-502, g, vmul24(100, f)) >> 9; }
__device__ __forceinline__ int c35n(int f, int g) { return vmad24(-284, g, vmul24(426, f)) >> 9; }

// dct.ml:114-149 (dct_col) / :151-187 (dct_row) share one butterfly.
// The butterfly after its first stage: from the sums a0..a3 and differences c0..c3.  LS = the
// sums still carry the +256 of two un-shifted pixels (see k_encode): only b0 and b1 see it.
template <bool LS>
__device__ __forceinline__ void fdct_tail(int a0, int a1, int a2, int a3, int c0, int c1, int c2, int c3, int &p0,
                                          int &p1, int &p2, int &p3, int &p4, int &p5, int &p6, int &p7) {
#if HVC_ENCODE_MULHI
    constexpr int C4S = 362 << 14;
    const int b0s = (a0 + a3) << 9, b1s = (a1 + a2) << 9, b2 = a1 - a2, b3 = a0 - a3;
    p0 = vmulhi24(C4S, LS ? b0s + b1s - (1024 << 9) : b0s + b1s);
    p4 = vmulhi24(C4S, b0s - b1s);   // c4 b0 (-b1): the level shift cancels
    p2 = c62(b2, b3);
    p6 = c62n(b3, b2);     // c62 b3 (-b2)
    const int c2s = c2 << 9;
    int b0 = vmulhi24(C4S, vmad24(-512, c1, c2s));   // c4 c2 (-c1)
    int b1 = vmulhi24(C4S, vmad24(512, c1, c2s));
#else
    int b0 = LS ? a0 + a3 - 512 : a0 + a3, b1 = LS ? a1 + a2 - 512 : a1 + a2, b2 = a1 - a2, b3 = a0 - a3;
    p0 = c4(b0, b1);
    p4 = c4m(b0, b1);      // c4 b0 (-b1)
    p2 = c62(b2, b3);
    p6 = c62n(b3, b2);     // c62 b3 (-b2)
    b0 = c4m(c2, c1);      // c4 c2 (-c1)
    b1 = c4(c2, c1);
#endif
    a0 = c0 + b0;
    a1 = c0 - b0;
    a2 = c3 - b1;
    a3 = c3 + b1;
    p1 = c71(a0, a3);
    p5 = c35(a1, a2);
    p3 = c35n(a2, a1);     // c35 a2 (-a1)
    p7 = c71n(a3, a0);     // c71 a3 (-a0)
}

// byte BYTE of a  +/-  byte BYTE of b: the unpacking of the pixel bytes rides in the SDWA source
// selectors of the first butterfly stage (zero-extended bytes, one instruction each)
template <int BYTE>
__device__ __forceinline__ int add_bytes(unsigned a, unsigned b) {
    int d;
    if (BYTE == 0) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_0" : "=v"(d) : "v"(a), "v"(b));
    if (BYTE == 1) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_1" : "=v"(d) : "v"(a), "v"(b));
    if (BYTE == 2) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:BYTE_2" : "=v"(d) : "v"(a), "v"(b));
    if (BYTE == 3) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:BYTE_3" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
template <int BYTE>
__device__ __forceinline__ int sub_bytes(unsigned a, unsigned b) {
    int d;
    if (BYTE == 0) asm("v_sub_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_0" : "=v"(d) : "v"(a), "v"(b));
    if (BYTE == 1) asm("v_sub_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_1" : "=v"(d) : "v"(a), "v"(b));
    if (BYTE == 2) asm("v_sub_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:BYTE_2" : "=v"(d) : "v"(a), "v"(b));
    if (BYTE == 3) asm("v_sub_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:BYTE_3" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

// Column pass (dct.ml:114-149) of column C from the eight packed pixel rows (two dwords each).
// level_shifted_input_block (encoder.ml:87) subtracts 128 from every pixel: differences of two
// pixels do not see it, sums carry +256, and past the first stage only b0 = a0 + a3 and b1 = a1 + a2
// use sums -- so the shift is two "- 512" (fdct_tail<true>) instead of 64 subtractions.
template <int C>
__device__ __forceinline__ void fdct_col_bytes(const unsigned (&px)[8][2], int (&v)[64]) {
    constexpr int D = C >> 2, B = C & 3;
    const int a0 = add_bytes<B>(px[0][D], px[7][D]), c3 = sub_bytes<B>(px[0][D], px[7][D]);
    const int a1 = add_bytes<B>(px[1][D], px[6][D]), c2 = sub_bytes<B>(px[1][D], px[6][D]);
    const int a2 = add_bytes<B>(px[2][D], px[5][D]), c1 = sub_bytes<B>(px[2][D], px[5][D]);
    const int a3 = add_bytes<B>(px[3][D], px[4][D]), c0 = sub_bytes<B>(px[3][D], px[4][D]);
    fdct_tail<true>(a0, a1, a2, a3, c0, c1, c2, c3, v[C], v[8 + C], v[16 + C], v[24 + C], v[32 + C], v[40 + C],
                    v[48 + C], v[56 + C]);
}

// Encoder.quant_and_scale (encoder.ml:98-101): trunc((f +- 2t) / (4t)) = f/(4t) rounded to the
// nearest integer, halves away from zero.  Evaluated as floor(f * r + 0.5) (v_cvt_rpi_i32_f32) with
// r = fl((1 + 2^-16) / (4t)): the exact quotients are multiples of 1/(4t), so a non-tie is at
// least 1/(8t) from a rounding boundary while the relative bias moves it by < 2^-16 * 2^11; a tie
// (k + 1/2) is pushed just past the boundary, away from zero, for either sign.  Exhaustively equal
// to the model for every t in 1..255 and |f| <= 2^15 (tests/test_quant_division.py).
__device__ __forceinline__ int quant1(int f, float r) {
    int q;
    const float x = (float)f * r;
    asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(q) : "v"(x));
    return q;
}

#ifndef HVC_ENCODE_LB
#define HVC_ENCODE_LB HVC_TILE
#endif
#ifdef HVC_ENCODE_WAVES /* experiments: -DHVC_ENCODE_WAVES=4 */
#define HVC_ENCODE_ATTR __attribute__((amdgpu_waves_per_eu(HVC_ENCODE_WAVES, HVC_ENCODE_WAVES)))
#else
#define HVC_ENCODE_ATTR
#endif
#ifndef HVC_ENCODE_NT_LOADS
// ... and the pixel rows arrive as non-temporal loads: a wave's row load is 512 contiguous bytes of ONE instruction, so there
// is nothing for the L1 to merge across instructions (unlike K1's eight 16-byte reads of a lane's 128-byte line), and the
// pixels are read once: +0.3 ... +0.8 points of the HBM peak, three alternations (profiles/r04t_k3_nt_loads.txt)
#define HVC_ENCODE_NT_LOADS 1
#endif
#ifndef HVC_ENCODE_NT
// the 1 KiB runs leave as non-temporal stores: +2.5 % (1.754 -> 1.712 ms per 256 4K frames, same box, alternating
// runs; profiles/r02e_ab.txt).  (For the lane-strided 16-byte pieces of the first version nt stores were 5x worse.)
#define HVC_ENCODE_NT 1
#endif
__global__ __launch_bounds__(HVC_ENCODE_LB) HVC_ENCODE_ATTR void k_encode(EncodeParams P) {
    BlockRef br;
    const int lane = threadIdx.x;
    unsigned wframe, wtile;
    xcd_work(P.xcd_map, P.xcd_magic, wframe, wtile);
    const bool active = locate(P, (int)wframe, (int)wtile, lane, br);
    const uint8_t *pix = P.pixels + br.pix_idx;
    // 8 rows x 8 B per lane; a wave's row loads are 512 contiguous bytes of a pixel row.
    unsigned px[8][2];
#pragma unroll
    for (int j = 0; j < 8; j++) {
#if HVC_ENCODE_NT_LOADS
        typedef unsigned u2l __attribute__((ext_vector_type(2)));
        const u2l w = __builtin_nontemporal_load(reinterpret_cast<const u2l *>(pix + (size_t)j * br.stride));
#else
        const uint2 w = *reinterpret_cast<const uint2 *>(pix + (size_t)j * br.stride);
#endif
        px[j][0] = w.x;
        px[j][1] = w.y;
    }
    // Dct.Chen.forward_8x8 (dct.ml:189-196): columns first, then rows
    int v[64];
#if HVC_TRAFFIC_ONLY
#pragma unroll
    for (int k = 0; k < 64; k++) v[k] = (int)((px[k >> 3][(k >> 2) & 1] >> (8 * (k & 3))) & 0xffu);
#else
    fdct_col_bytes<0>(px, v);
    fdct_col_bytes<1>(px, v);
    fdct_col_bytes<2>(px, v);
    fdct_col_bytes<3>(px, v);
    fdct_col_bytes<4>(px, v);
    fdct_col_bytes<5>(px, v);
    fdct_col_bytes<6>(px, v);
    fdct_col_bytes<7>(px, v);
#pragma unroll
    for (int r = 0; r < 8; r++) {
        int *p = v + r * 8;
        fdct_tail<false>(p[0] + p[7], p[1] + p[6], p[2] + p[5], p[3] + p[4], p[3] - p[4], p[2] - p[5], p[1] - p[6],
                         p[0] - p[7], p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7]);
    }
#endif
    // Encoder.quant (encoder.ml:103-108): quant[zz] = quant_and_scale fdct[ZI[zz]] table[zz]
    const float *__restrict__ qr = P.qrcp + br.qtab * 64;
    // A lane holds its block's 128 output bytes; storing them as eight 16-byte pieces 128 bytes apart
    // across lanes makes every piece its own L2 write request (TCC requests 8x, 4.5 TB/s ceiling in
    // tools/ubench/store_ubench; non-temporal: 5x slower still).  So each wave transposes its 8 KiB
    // through LDS -- XOR-swizzled 16-byte slots, conflict-free for both the ds_write_b128 (lane l
    // writes slot 8l + (c ^ (l & 7))) and the ds_read_b128 -- and stores whole 1 KiB runs
    // (lane i of store j writes byte 1024 j + 16 i of the wave's coefficient run).
    typedef unsigned u4v __attribute__((ext_vector_type(4)));
    __shared__ u4v lds[HVC_TILE / 64][512];
    const int wv = lane >> 6, l = lane & 63;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        unsigned w[4];
#pragma unroll
        for (int h = 0; h < 4; h++) {
            const int k = j * 8 + h * 2;
#if HVC_TRAFFIC_ONLY
            const int lo = v[ZI[k]], hi = v[ZI[k + 1]];
            (void)qr;
#else
#if HVC_ENCODE_QMAGIC
            // quant_and_scale as ONE fused multiply-add per coefficient: float(f) * r + 1.5 * 2^23 rounds (to nearest, once, at
            // an ulp of 1) to 1.5 * 2^23 + q, whose low 16 bits ARE q in two's complement -- no v_cvt_rpi, and the int16 pair is
            // one v_perm_b32 of the two bit patterns.  The exact product f * r is never a tie and its nearest integer is the
            // model's quotient for every t in 1..255 and |f| <= 2^15 (tests/test_quant_division.py, exhaustive).
#if HVC_ENCODE_QMAGIC == 2
            typedef float f2v __attribute__((ext_vector_type(2)));
            const f2v xf = {(float)v[ZI[k]], (float)v[ZI[k + 1]]}, rf = {qr[k], qr[k + 1]}, mf = {12582912.f, 12582912.f};
            const f2v pr = __builtin_elementwise_fma(xf, rf, mf);   // v_pk_fma_f32
            const float plo = pr.x, phi = pr.y;
#else
            const float plo = __builtin_fmaf((float)v[ZI[k]], qr[k], 12582912.f);
            const float phi = __builtin_fmaf((float)v[ZI[k + 1]], qr[k + 1], 12582912.f);
#endif
            w[h] = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, phi), __builtin_bit_cast(unsigned, plo), 0x05040100u);
#elif HVC_ENCODE_MULHI
            // the two products of a pair as one v_pk_mul_f32 (the same IEEE single products as two v_mul_f32)
            typedef float f2v __attribute__((ext_vector_type(2)));
            const f2v xf = {(float)v[ZI[k]], (float)v[ZI[k + 1]]}, rf = {qr[k], qr[k + 1]};
            const f2v pr = xf * rf;
            int lo, hi;
            asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(lo) : "v"(pr.x));
            asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(hi) : "v"(pr.y));
#else
            const int lo = quant1(v[ZI[k]], qr[k]);
            const int hi = quant1(v[ZI[k + 1]], qr[k + 1]);
#endif
#endif
#if HVC_TRAFFIC_ONLY || !HVC_ENCODE_QMAGIC
            w[h] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pk_i16(lo, hi)); // |q| <= 2^13: no saturation
#endif
        }
        const u4v t = {w[0], w[1], w[2], w[3]};
        lds[wv][l * 8 + (j ^ (l & 7))] = t;
    }
    // (wave-private LDS region: the wave's own ds ops are ordered, no barrier needed)
    const int wave_b0 = br.tile_b0 + (lane & ~63);
    u4v *dst = reinterpret_cast<u4v *>(P.coefs + br.plane_coef_idx + (size_t)wave_b0 * 64);
    (void)active;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int blk = 8 * j + (l >> 3), ch = l & 7;
        const u4v t = lds[wv][blk * 8 + (ch ^ (blk & 7))];
        if (wave_b0 + blk < br.nblk) {
#if HVC_ENCODE_NT
            __builtin_nontemporal_store(t, dst + j * 64 + l);
#else
            dst[j * 64 + l] = t;
#endif
        }
    }
}

// ---------------------------------------------------------------------------
// K2: 4:2:0 -> 4:4:4 chroma upsample, tools/src/planar_444.ml:82-103
// (supersample_hv2 for every row, :122-131).  One thread per 4 source pixels of
// a source row: writes 8 + 8 destination bytes (rows 2r and 2r+1).

template <bool VEC>
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
    uint8_t *d1 = dst + (size_t)(2 * row) * P.dst_stride + 2 * c0;
    uint8_t *d2 = d1 + P.dst_stride;
    if (VEC) {
        // aligned fast path (cw % 4 == 0, strides and bases aligned; chosen on the host): one dword
        // per source row, one 8-byte store per destination row, 256 B / 512 B contiguous per wave
        const unsigned wa = *reinterpret_cast<const unsigned *>(s1 + c0);
        const unsigned wb = *reinterpret_cast<const unsigned *>(s2 + c0);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            a[i] = (wa >> (8 * i)) & 0xffu;
            b[i] = (wb >> (8 * i)) & 0xffu;
        }
        const bool last = c0 + 4 >= P.cw; // last column replicates (:98-102)
        a[4] = last ? a[3] : s1[c0 + 4];
        b[4] = last ? b[3] : s2[c0 + 4];
        unsigned o1[8], o2[8];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            o1[2 * i] = a[i];
            o1[2 * i + 1] = avg2u(a[i], a[i + 1]);
            o2[2 * i] = avg2u(a[i], b[i]);
            o2[2 * i + 1] = avg4u(a[i], a[i + 1], b[i], b[i + 1]);
        }
        *reinterpret_cast<uint2 *>(d1) = make_uint2(o1[0] | (o1[1] << 8) | (o1[2] << 16) | (o1[3] << 24),
                                                    o1[4] | (o1[5] << 8) | (o1[6] << 16) | (o1[7] << 24));
        *reinterpret_cast<uint2 *>(d2) = make_uint2(o2[0] | (o2[1] << 8) | (o2[2] << 16) | (o2[3] << 24),
                                                    o2[4] | (o2[5] << 8) | (o2[6] << 16) | (o2[7] << 24));
        return;
    }
#pragma unroll
    for (int i = 0; i < 5; i++) {
        int c = min(c0 + i, P.cw - 1); // last column replicates (:98-102)
        a[i] = s1[c];
        b[i] = s2[c];
    }
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
// K2, wide form (cw % 8 == 0, 8 / 16-byte aligned rows): one thread = 8 source samples of one source row ->
// two 16-byte output rows, with the packed-byte arithmetic of k_decode_444's epilogue (rowq / emit_rows444:
// v_lerp_u8 averages, v_perm interleave, non-temporal 16-byte stores, 1 KiB contiguous per wave instruction).
__global__ __launch_bounds__(256) void k_upsample420_x8(UpsampleParams P) {
    const int groups = P.cw >> 3;
    const long long total = (long long)groups * P.ch;
    unsigned wplane, wtile;
    xcd_work(P.xcd_map, P.xcd_magic, wplane, wtile);
    const long long t = (long long)wtile * 256 + threadIdx.x;
    if (t >= total) return;
    const int row = (int)(t / groups), g = (int)(t % groups);
    const uint8_t *src = P.src + (size_t)wplane * P.src_ps;
    uint8_t *dst = P.dst + (size_t)wplane * P.dst_ps;
    const int row2 = min(P.ch - 1, row + 1); // :86
    const uint8_t *s1 = src + (size_t)row * P.src_stride + 8 * g;
    const uint8_t *s2 = src + (size_t)row2 * P.src_stride + 8 * g;
    const uint2 a = *reinterpret_cast<const uint2 *>(s1), b = *reinterpret_cast<const uint2 *>(s2);
    const bool last = g == groups - 1; // the w - 1 column replicates (:97-102): b = a there
    const unsigned an = last ? 0u : s1[8], bn = last ? 0u : s2[8];
    const unsigned m1 = last ? 0xff000000u : 0u;
    const RowQ cur = rowq<0>(a.x, a.y, an, 0u, m1), nxt = rowq<0>(b.x, b.y, bn, 0u, m1);
    emit_rows444<true>(dst + (size_t)(2 * row) * P.dst_stride, P.dst_stride, 16 * g, cur, nxt, false);
}

// ---------------------------------------------------------------------------
// Encoder.recon's error plane (encoder.ml:119-125): error.(i) = abs (recon.(i) - input_pixels.(i)), block per
// lane with K3's load shape (8 rows x 8 B; a wave's row accesses are 512 contiguous bytes).  `pixels` of the
// EncodeParams = the encoder's input, `recon` = what the block stage rebuilt from the quantised coefficients
// (K1's output: max 0 (min 255 (idct + 128)) is K1's clip + level shift), both in the pixel-record layout.
__device__ __forceinline__ unsigned absdiff_u8x4(unsigned a, unsigned b) {
    unsigned r = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int d = (int)((a >> (8 * k)) & 0xffu) - (int)((b >> (8 * k)) & 0xffu);
        r |= (unsigned)(d < 0 ? -d : d) << (8 * k);
    }
    return r;
}
__global__ __launch_bounds__(HVC_TILE) void k_abs_error(EncodeParams P, const uint8_t *recon, uint8_t *error) {
    BlockRef br;
    if (!locate(P, blockIdx.y, blockIdx.x, threadIdx.x, br)) return;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const size_t at = br.pix_idx + (size_t)j * br.stride;
        const uint2 a = *reinterpret_cast<const uint2 *>(P.pixels + at), b = *reinterpret_cast<const uint2 *>(recon + at);
        *reinterpret_cast<uint2 *>(error + at) = make_uint2(absdiff_u8x4(b.x, a.x), absdiff_u8x4(b.y, a.y));
    }
}
hipError_t launch_abs_error(const EncodeParams &P, const uint8_t *recon, uint8_t *error, hipStream_t s) {
    if (P.n_frames <= 0 || P.tiles_per_frame <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_abs_error, dim3((unsigned)P.tiles_per_frame, (unsigned)P.n_frames, 1), dim3(HVC_TILE), 0, s, P, recon, error);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// K5: checksum of byte records on the device, so that a benchmark can say WHAT it produced without
// bringing the frames back (no counterpart in the reference; SURVEY.md section 2 "K5").
//   sum[r] = SUM_i (byte_i + 1) * ((2 i + 1) * HVC_CHECKSUM_MUL)   (mod 2^64), i = byte index inside the record
// -- position-weighted (odd weights: a changed byte always changes the sum, swapped bytes do unless equal) and
// a plain sum, hence order-free: any split into partial sums gives the same 64 bits, on the host too
// (numpy uint64 arithmetic wraps the same way).  One pass over the data, 16 bytes per lane per
// iteration, one 64-bit atomic per wavefront.
__global__ __launch_bounds__(256) void k_checksum(const uint8_t *data, size_t record_bytes, size_t record_stride,
                                                  unsigned long long *sums) {
    const uint8_t *rec = data + (size_t)blockIdx.y * record_stride;
    const size_t vec = ((uintptr_t)rec & 15) == 0 ? record_bytes / 16 : 0; // 16-byte pieces (aligned records only)
    unsigned long long acc = 0;
    for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < vec; v += (size_t)gridDim.x * 256) {
        const uint4 q = reinterpret_cast<const uint4 *>(rec)[v];
        const unsigned w[4] = {q.x, q.y, q.z, q.w};
        // SUM (b_k + 1) * (2 (i0 + k) + 1) M = M * [ (2 i0 + 1) * S0 + 2 * S1 ],  S0 = SUM (b_k + 1), S1 = SUM k (b_k + 1)
        unsigned s0 = 0, s1 = 0;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const unsigned b = ((w[k >> 2] >> (8 * (k & 3))) & 0xffu) + 1u;
            s0 += b;
            s1 += (unsigned)k * b;
        }
        acc += (2ull * (16ull * v) + 1ull) * s0 + 2ull * s1;
    }
    if (blockIdx.x == 0) // the bytes the 16-byte pieces do not cover
        for (size_t i = vec * 16 + threadIdx.x; i < record_bytes; i += 256) acc += (2ull * i + 1ull) * ((unsigned long long)rec[i] + 1ull);
    acc *= HVC_CHECKSUM_MUL;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0 && acc) atomicAdd(&sums[blockIdx.y], acc);
}

hipError_t launch_checksum(const uint8_t *data, size_t record_bytes, size_t record_stride, int n_records,
                           unsigned long long *sums, hipStream_t s) {
    if (n_records <= 0) return hipSuccess;
    hipError_t e = hipMemsetAsync(sums, 0, (size_t)n_records * sizeof(unsigned long long), s);
    if (e != hipSuccess) return e;
    if (record_bytes == 0) return hipSuccess;
    // enough workgroups to fill the chip however few records there are; a grid-stride loop takes the rest
    const size_t pieces = (record_bytes / 16 + 255) / 256;
    size_t gx = pieces < 1 ? 1 : pieces;
    const size_t want = (size_t)(4096 + n_records - 1) / (size_t)n_records;
    if (gx > want) gx = want;
    hipLaunchKernelGGL(k_checksum, dim3((unsigned)gx, (unsigned)n_records, 1), dim3(256), 0, s, data, record_bytes,
                       record_stride, sums);
    return hipGetLastError();
}


hipError_t launch_decode(const DecodeParams &P, hipStream_t s, hipEvent_t k0, hipEvent_t k1) {
    if (P.n_frames <= 0 || P.tiles_per_frame <= 0) return hipSuccess;
    hipError_t e;
    dim3 grid((unsigned)P.tiles_per_frame, (unsigned)P.n_frames, 1);
    if (k0 && (e = hipEventRecord(k0, s)) != hipSuccess) return e;
    // HVC_DECODE_KERNEL=v2 / q16 selects the unpacked int32 kernel / the quarter-wavefront kernel
    // (A/B measurements only; hvc_set_decode_kernel is the API)
    static const int env_sel = [] {
        const char *v = getenv("HVC_DECODE_KERNEL");
        return !v ? 0 : (v[0] == 'v' && v[1] == '2') ? 1 : (v[0] == 'q') ? 3 : 0;
    }();
    // (the A/B alternates read the DC from the record: with a compact DC array the environment's choice does not apply,
    // and hvc_capi.hip never combines an explicit selection with one)
    const int sel = P.kernel_sel ? P.kernel_sel : P.dc_plane ? 0 : env_sel;
    if (P.dc_plane && sel != 0) return hipErrorInvalidValue;
    if (sel == 3)
        hipLaunchKernelGGL(k_decode_q16, grid, dim3(HVC_TILE), 0, s, P);
    else if (sel == 1)
        hipLaunchKernelGGL(k_decode_fast, grid, dim3(HVC_TILE), 0, s, P);
    else {
        static const unsigned pad = [] { const char *v = getenv("HVC_DEC_LDS_PAD"); return v ? (unsigned)atoi(v) : 0u; }(); // experiments: unused LDS caps the workgroups per CU
        DecodeParams Q = P;
        Q.xcd_map = xcd_map_for(grid.x, grid.y, Q.xcd_magic); // (xcd_work: every XCD takes runs of consecutive tiles)
        if (P.dc_plane)
            hipLaunchKernelGGL(k_decode_packed<true>, grid, dim3(HVC_TILE), pad, s, Q);
        else
            hipLaunchKernelGGL(k_decode_packed<false>, grid, dim3(HVC_TILE), pad, s, Q);
    }
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (k1 && (e = hipEventRecord(k1, s)) != hipSuccess) return e;
    // Fixed small grid; every thread strides over the (normally empty) list and
    // exits as soon as its index passes *fix_count.
    hipLaunchKernelGGL(k_decode_wide, dim3(HVC_FIXUP_WGS), dim3(HVC_FIXUP_LANES), 0, s, P, P.fix_count, P.fix_list, (const long long *)nullptr);
    return hipGetLastError();
}

// Blocks whose true DC does not fit the record (hvc_hdec.h WideDc), after the batch's normal launches: recomputed in
// int64 with that DC.  ids = fix-list ids of P's geometry, dcs parallel, *count entries (all device memory).
hipError_t launch_decode_dcfix(const DecodeParams &P, const unsigned *count, const unsigned *ids, const long long *dcs,
                               hipStream_t s) {
    DecodeParams Q = P;
    Q.fix_count_next = nullptr; // (not part of the launches' counter ping-pong)
    Q.wide_total = nullptr;
    hipLaunchKernelGGL(k_decode_wide, dim3(64), dim3(64), 0, s, Q, count, ids, dcs);
    return hipGetLastError();
}

hipError_t launch_decode_444_dcfix(const Decode444Params &P, const unsigned *count, const unsigned *ids, const long long *dcs,
                                   unsigned n_host, hipStream_t s) {
    Decode444Params Q = P;
    Q.fix_count_next = nullptr;
    Q.wide_total = nullptr;
    hipLaunchKernelGGL(k_decode_wide_444, dim3(64), dim3(64), 0, s, Q, count, ids, 0ull, dcs);
    // the interpolated samples around the rewritten chroma blocks (9 x 9 source samples each); the seam rows are
    // rebuilt once more on the way, from the same source samples
    const size_t W = (size_t)P.width;
    const bool aligned = (W % 16 == 0) && (P.out_fs % 16 == 0) && ((uintptr_t)P.out % 16 == 0);
    const long long hw = aligned ? P.pl[1].aw / 8 : P.pl[1].aw;
    const long long seams = 2 * (long long)(P.c_tiles_y - 1) * hw;
    const long long want = seams > (long long)n_host * 81 ? seams : (long long)n_host * 81;
    const dim3 rgrid((unsigned)((want + 255) / 256) ? (unsigned)((want + 255) / 256) : 1u, (unsigned)P.n_frames, 1);
    if (aligned)
        hipLaunchKernelGGL(k_reinterp_444<true>, rgrid, dim3(256), 0, s, Q, count, ids, 0);
    else
        hipLaunchKernelGGL(k_reinterp_444<false>, rgrid, dim3(256), 0, s, Q, count, ids, 0);
    return hipGetLastError();
}

hipError_t launch_decode_wide_only(const DecodeParams &P, hipStream_t s, hipEvent_t k0, hipEvent_t k1) {
    if (P.n_frames <= 0 || P.tiles_per_frame <= 0) return hipSuccess;
    hipError_t e;
    const dim3 grid((unsigned)P.tiles_per_frame, (unsigned)P.n_frames, 1);
    DecodeParams Q = P;
    Q.xcd_map = xcd_map_for(grid.x, grid.y, Q.xcd_magic);
    if (k0 && (e = hipEventRecord(k0, s)) != hipSuccess) return e;
    hipLaunchKernelGGL(k_decode_wide_all, grid, dim3(HVC_TILE), 0, s, Q);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    if (k1 && (e = hipEventRecord(k1, s)) != hipSuccess) return e;
    return hipSuccess;
}

void plan_decode_444(Decode444Params &P, bool aligned) {
    const int cbw = P.pl[1].cbw;
    // one chroma tile across the whole row where 256 blocks reach (the byte-store form for odd sizes keeps the small tile)
    P.nw = !aligned || cbw <= HVC_444_TILE_BW ? 1 : cbw <= 2 * HVC_444_TILE_BW ? 2 : 4;
    const int wgs = HVC_TILE * P.nw, tw = HVC_444_TILE_BW * P.nw;
    P.y_tiles = (P.pl[0].cbw * P.pl[0].cbh + wgs - 1) / wgs;
    P.y_magic = (unsigned)(((1ull << 32) + P.pl[0].cbw - 1) / P.pl[0].cbw);
    P.c_tiles_x = cbw <= tw ? 1 : (cbw - 1 + (tw - 1) - 1) / (tw - 1);
    P.c_tiles_y = (P.pl[1].cbh + HVC_444_TILE_BH - 1) / HVC_444_TILE_BH;
    P.c_magic = (unsigned)(((1ull << 32) + P.c_tiles_x - 1) / P.c_tiles_x);
    P.tiles_per_frame = P.y_tiles + 2 * P.c_tiles_x * P.c_tiles_y;
}

template <bool ALIGNED, int NW>
static void launch_444_kernel(const Decode444Params &Q, dim3 grid, unsigned pad, hipStream_t s) {
    if (Q.dc_plane)
        hipLaunchKernelGGL((k_decode_444<ALIGNED, true, NW>), grid, dim3(HVC_TILE * NW), pad, s, Q);
    else
        hipLaunchKernelGGL((k_decode_444<ALIGNED, false, NW>), grid, dim3(HVC_TILE * NW), pad, s, Q);
}

hipError_t launch_decode_444(const Decode444Params &P, bool wide_only, hipStream_t s, hipEvent_t k0, hipEvent_t k1) {
    if (P.n_frames <= 0 || P.tiles_per_frame <= 0) return hipSuccess;
    if (P.nw != 1 && P.nw != 2 && P.nw != 4) return hipErrorInvalidValue;
    const dim3 grid((unsigned)P.tiles_per_frame, (unsigned)P.n_frames, 1);
    const long long per_plane = (long long)P.pl[1].aw * P.pl[1].ah;
    if (wide_only) {
        const unsigned long long total = (unsigned long long)P.n_frames * P.tiles_per_frame * HVC_TILE * P.nw;
        if (k0) (void)hipEventRecord(k0, s);
        hipLaunchKernelGGL(k_decode_wide_444, dim3(4096), dim3(64), 0, s, P, (const unsigned *)nullptr,
                           (const unsigned *)nullptr, total, (const long long *)nullptr);
        if (k1) (void)hipEventRecord(k1, s);
        hipLaunchKernelGGL(k_reinterp_444<false>, dim3((unsigned)((2 * per_plane + 255) / 256), (unsigned)P.n_frames, 1),
                           dim3(256), 0, s, P, (const unsigned *)P.fix_count, (const unsigned *)P.fix_list, 1);
        return hipGetLastError();
    }
    const size_t W = (size_t)P.width;
    const bool aligned = (W % 16 == 0) && (P.out_fs % 16 == 0) && ((uintptr_t)P.out % 16 == 0);
    if (k0) (void)hipEventRecord(k0, s);
    // HVC_444_ONLY=luma|chroma (measurements): the other half of the tiles returns at once
    static const int only = [] { const char *v = getenv("HVC_444_ONLY"); return !v ? 0 : v[0] == 'l' ? 2 : v[0] == 'c' ? 1 : 0; }();
    Decode444Params Q = P;
    Q.skip = only;
    Q.xcd_map = xcd_map_for((unsigned)(P.tiles_per_frame - P.tile0), (unsigned)P.n_frames, Q.xcd_magic, true);
    // HVC_444_LDS_PAD=bytes (experiments): dynamic LDS nobody uses, to hold the kernel to fewer workgroups per CU
    static const unsigned pad = [] { const char *v = getenv("HVC_444_LDS_PAD"); return v ? (unsigned)atoi(v) : 0u; }();
    if (!aligned && P.nw != 1) return hipErrorInvalidValue; // (plan_decode_444 was told otherwise)
    // P.tile0 = P.y_tiles: the chroma tiles alone (the luma planes went through k_decode_packed: hvc_capi.hip, decode_frames_yuv444_impl)
    const dim3 cgrid((unsigned)(P.tiles_per_frame - P.tile0), (unsigned)P.n_frames, 1);
    if (!aligned)
        launch_444_kernel<false, 1>(Q, cgrid, 0, s);
    else if (P.nw == 1)
        launch_444_kernel<true, 1>(Q, cgrid, pad, s);
    else if (P.nw == 2)
        launch_444_kernel<true, 2>(Q, cgrid, pad, s);
    else
        launch_444_kernel<true, 4>(Q, cgrid, pad, s);
    if (k1) (void)hipEventRecord(k1, s);
    hipLaunchKernelGGL(k_decode_wide_444, dim3(256), dim3(64), 0, s, P, (const unsigned *)P.fix_count,
                       (const unsigned *)P.fix_list, 0ull, (const long long *)nullptr);
    const long long hw = aligned ? P.pl[1].aw / 8 : P.pl[1].aw;
    const long long seams = 2 * (long long)(P.c_tiles_y - 1) * hw;
    const dim3 rgrid((unsigned)((seams + 255) / 256) ? (unsigned)((seams + 255) / 256) : 1u, (unsigned)P.n_frames, 1);
    if (aligned)
        hipLaunchKernelGGL(k_reinterp_444<true>, rgrid, dim3(256), 0, s, P, (const unsigned *)P.fix_count,
                           (const unsigned *)P.fix_list, 0);
    else
        hipLaunchKernelGGL(k_reinterp_444<false>, rgrid, dim3(256), 0, s, P, (const unsigned *)P.fix_count,
                           (const unsigned *)P.fix_list, 0);
    return hipGetLastError();
}

hipError_t launch_encode(const EncodeParams &P, hipStream_t s, hipEvent_t k0, hipEvent_t k1) {
    if (P.n_frames <= 0 || P.tiles_per_frame <= 0) return hipSuccess;
    hipError_t e;
    dim3 grid((unsigned)P.tiles_per_frame, (unsigned)P.n_frames, 1);
    if (k0 && (e = hipEventRecord(k0, s)) != hipSuccess) return e;
    static const unsigned pad = [] { const char *v = getenv("HVC_ENC_LDS_PAD"); return v ? (unsigned)atoi(v) : 0u; }(); // experiments
    EncodeParams Q = P;
    Q.xcd_map = xcd_map_for(grid.x, grid.y, Q.xcd_magic);
    hipLaunchKernelGGL(k_encode, grid, dim3(HVC_TILE), pad, s, Q);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (k1 && (e = hipEventRecord(k1, s)) != hipSuccess) return e;
    return hipSuccess;
}

hipError_t launch_upsample420(const UpsampleParams &P, hipStream_t s) {
    if (P.n_planes <= 0 || P.cw <= 0 || P.ch <= 0) return hipSuccess;
    long long total = (long long)((P.cw + 3) >> 2) * P.ch;
    dim3 grid((unsigned)((total + 255) / 256), (unsigned)P.n_planes, 1);
    const bool vec = (P.cw % 4 == 0) && (P.src_stride % 4 == 0) && (P.dst_stride % 8 == 0) && (P.src_ps % 4 == 0) &&
                     (P.dst_ps % 8 == 0) && ((uintptr_t)P.src % 4 == 0) && ((uintptr_t)P.dst % 8 == 0);
    const bool x8 = (P.cw % 8 == 0) && (P.src_stride % 8 == 0) && (P.dst_stride % 16 == 0) && (P.src_ps % 8 == 0) &&
                    (P.dst_ps % 16 == 0) && ((uintptr_t)P.src % 8 == 0) && ((uintptr_t)P.dst % 16 == 0);
    if (x8) {
        const long long n8 = (long long)(P.cw >> 3) * P.ch;
        UpsampleParams Q = P;
        Q.xcd_map = xcd_map_for((unsigned)((n8 + 255) / 256), (unsigned)P.n_planes, Q.xcd_magic);
        hipLaunchKernelGGL(k_upsample420_x8, dim3((unsigned)((n8 + 255) / 256), (unsigned)P.n_planes, 1), dim3(256), 0,
                           s, Q);
    } else if (vec)
        hipLaunchKernelGGL(k_upsample420<true>, grid, dim3(256), 0, s, P);
    else
        hipLaunchKernelGGL(k_upsample420<false>, grid, dim3(256), 0, s, P);
    return hipGetLastError();
}

} // namespace hvc
