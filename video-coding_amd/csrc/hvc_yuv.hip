// hvc_yuv.hip -- the step after the path, the other half: the reference's `oyuv convert`
// (tools/src/oconv.ml) on the GPU.  Per frame Oconv.main reads the input format into a 4:4:4 frame, crops / offsets /
// edge-extends it to the output size (Yuv.crop) and writes the output format:
//     planar 4:2:0 <-> 4:4:4   Planar_444.supersample_hv2 / subsample_hv2     tools/src/planar_444.ml:69-131
//     planar 4:2:2 <-> 4:4:4   Planar_444.supersample_h2 / subsample_h2       tools/src/planar_444.ml:18-67
//     packed 4:2:2 <-> planar  Packed_422.convert_to_planar / _from_planar    tools/src/packed_422.ml:10-44
//     crop                     Yuv.crop (clamped source coordinates)           tools/src/yuv.ml:42-62
// (supersample_hv2 is K2, csrc/hvc_kernels.hip.)  All of it is byte traffic: one lane moves 8 destination (sub-sampling)
// or 8 source (super-sampling) samples with packed-byte arithmetic; a wave's loads and stores are contiguous runs of a row.
#include "hvc_ctx.h"

namespace {

struct PlaneOp {
    const uint8_t *src;
    uint8_t *dst;
    int sw, sh, dw, dh;   // source / destination plane size in samples
    int x_pos, y_pos;     // Yuv.crop only
    int vec;              // bases, strides and plane strides allow the 16 / 8-byte forms
    int xcd_map;          // workgroup -> (plane, piece of the plane) by hvc::xcd_work (runs of pieces per XCD), 0 = as dispatched
    unsigned xcd_magic, pad;
    size_t src_stride, dst_stride, src_ps, dst_ps;
};

typedef unsigned u4v __attribute__((ext_vector_type(4)));
typedef unsigned u2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pairsum(unsigned x) { return (x & 0x00ff00ffu) + ((x >> 8) & 0x00ff00ffu); } // (b0 + b1, b2 + b3) in 16-bit halves
__device__ __forceinline__ unsigned pack_even(unsigned lo, unsigned hi) { // bytes 0 and 2 of lo, then of hi
    return (lo & 0xffu) | ((lo >> 8) & 0xff00u) | ((hi & 0xffu) << 16) | ((hi & 0xff0000u) << 8);
}
__device__ __forceinline__ unsigned avg2x4(unsigned a, unsigned b) { // (a + b + 1) >> 1 on four bytes (planar_444.ml:4-8)
    return (a | b) - (((a ^ b) >> 1) & 0x7f7f7f7fu);
}

// Planar_444.subsample_hv2 (planar_444.ml:69-80): dst[col, row] = avg4 of the 2 x 2 source samples; dst = (sw / 2) x (sh / 2)
__global__ __launch_bounds__(256) void k_subsample420(PlaneOp P) {
    const unsigned groups = (unsigned)(P.dw + 7) >> 3;
    unsigned wplane, wpiece;
    hvc::xcd_work(P.xcd_map, P.xcd_magic, wplane, wpiece);
    const unsigned t = wpiece * 256u + threadIdx.x;
    if (t >= groups * (unsigned)P.dh) return;
    const unsigned row = t / groups, g = t - row * groups;
    const uint8_t *s0 = P.src + (size_t)wplane * P.src_ps + (size_t)(2 * row) * P.src_stride + 16 * g;
    const uint8_t *s1 = s0 + P.src_stride;
    uint8_t *d = P.dst + (size_t)wplane * P.dst_ps + (size_t)row * P.dst_stride + 8 * g;
    if (P.vec && (int)(8 * g + 8) <= P.dw) {
        const u4v a = __builtin_nontemporal_load(reinterpret_cast<const u4v *>(s0));
        const u4v b = __builtin_nontemporal_load(reinterpret_cast<const u4v *>(s1));
        const unsigned r0 = ((pairsum(a.x) + pairsum(b.x) + 0x00020002u) >> 2) & 0x00ff00ffu; // avg4 (planar_444.ml:10-16)
        const unsigned r1 = ((pairsum(a.y) + pairsum(b.y) + 0x00020002u) >> 2) & 0x00ff00ffu;
        const unsigned r2 = ((pairsum(a.z) + pairsum(b.z) + 0x00020002u) >> 2) & 0x00ff00ffu;
        const unsigned r3 = ((pairsum(a.w) + pairsum(b.w) + 0x00020002u) >> 2) & 0x00ff00ffu;
        const u2v o = {pack_even(r0, r1), pack_even(r2, r3)};
        __builtin_nontemporal_store(o, reinterpret_cast<u2v *>(d));
        return;
    }
    for (int i = 0; i < 8 && (int)(8 * g) + i < P.dw; i++)
        d[i] = (uint8_t)(((unsigned)s0[2 * i] + s0[2 * i + 1] + s1[2 * i] + s1[2 * i + 1] + 2u) >> 2);
}

// Planar_444.subsample_h2 (planar_444.ml:18-23): dst[col, row] = avg2 src[2 col, row] src[2 col + 1, row]; dst = (sw / 2) x sh
__global__ __launch_bounds__(256) void k_subsample422(PlaneOp P) {
    const unsigned groups = (unsigned)(P.dw + 7) >> 3;
    unsigned wplane, wpiece;
    hvc::xcd_work(P.xcd_map, P.xcd_magic, wplane, wpiece);
    const unsigned t = wpiece * 256u + threadIdx.x;
    if (t >= groups * (unsigned)P.dh) return;
    const unsigned row = t / groups, g = t - row * groups;
    const uint8_t *s0 = P.src + (size_t)wplane * P.src_ps + (size_t)row * P.src_stride + 16 * g;
    uint8_t *d = P.dst + (size_t)wplane * P.dst_ps + (size_t)row * P.dst_stride + 8 * g;
    if (P.vec && (int)(8 * g + 8) <= P.dw) {
        const u4v a = __builtin_nontemporal_load(reinterpret_cast<const u4v *>(s0));
        const unsigned r0 = ((pairsum(a.x) + 0x00010001u) >> 1) & 0x00ff00ffu, r1 = ((pairsum(a.y) + 0x00010001u) >> 1) & 0x00ff00ffu;
        const unsigned r2 = ((pairsum(a.z) + 0x00010001u) >> 1) & 0x00ff00ffu, r3 = ((pairsum(a.w) + 0x00010001u) >> 1) & 0x00ff00ffu;
        const u2v o = {pack_even(r0, r1), pack_even(r2, r3)};
        __builtin_nontemporal_store(o, reinterpret_cast<u2v *>(d));
        return;
    }
    for (int i = 0; i < 8 && (int)(8 * g) + i < P.dw; i++) d[i] = (uint8_t)(((unsigned)s0[2 * i] + s0[2 * i + 1] + 1u) >> 1);
}

// Planar_444.supersample_h2 (planar_444.ml:25-33): dst[2 col] = src[col], dst[2 col + 1] = avg2 src[col] src[col + 1]; the last
// column twice (avg2 a a = a: the right neighbour of the last column is the column itself); dst = (2 sw) x sh
__global__ __launch_bounds__(256) void k_upsample422(PlaneOp P) {
    const unsigned groups = (unsigned)(P.sw + 7) >> 3;
    unsigned wplane, wpiece;
    hvc::xcd_work(P.xcd_map, P.xcd_magic, wplane, wpiece);
    const unsigned t = wpiece * 256u + threadIdx.x;
    if (t >= groups * (unsigned)P.sh) return;
    const unsigned row = t / groups, g = t - row * groups;
    const uint8_t *s = P.src + (size_t)wplane * P.src_ps + (size_t)row * P.src_stride + 8 * g;
    uint8_t *d = P.dst + (size_t)wplane * P.dst_ps + (size_t)row * P.dst_stride + 16 * g;
    if (P.vec && (int)(8 * g + 8) <= P.sw) {
        const u2v a = __builtin_nontemporal_load(reinterpret_cast<const u2v *>(s));
        const unsigned next = (int)(8 * g + 8) < P.sw ? s[8] : (a.y >> 24);
        const unsigned n0 = (a.x >> 8) | (a.y << 24), n1 = (a.y >> 8) | (next << 24); // every sample's right neighbour
        const unsigned v0 = avg2x4(a.x, n0), v1 = avg2x4(a.y, n1);
        u4v o;
        o.x = (a.x & 0xffu) | ((v0 & 0xffu) << 8) | ((a.x & 0xff00u) << 8) | ((v0 & 0xff00u) << 16);
        o.y = ((a.x >> 16) & 0xffu) | ((v0 >> 8) & 0xff00u) | ((a.x >> 8) & 0xff0000u) | (v0 & 0xff000000u);
        o.z = (a.y & 0xffu) | ((v1 & 0xffu) << 8) | ((a.y & 0xff00u) << 8) | ((v1 & 0xff00u) << 16);
        o.w = ((a.y >> 16) & 0xffu) | ((v1 >> 8) & 0xff00u) | ((a.y >> 8) & 0xff0000u) | (v1 & 0xff000000u);
        __builtin_nontemporal_store(o, reinterpret_cast<u4v *>(d));
        return;
    }
    for (int i = 0; i < 8 && (int)(8 * g) + i < P.sw; i++) {
        const int c = (int)(8 * g) + i, c2 = c + 1 < P.sw ? c + 1 : P.sw - 1;
        const uint8_t *r = s - 8 * g;
        d[2 * i] = r[c];
        d[2 * i + 1] = (uint8_t)(((unsigned)r[c] + r[c2] + 1u) >> 1);
    }
}

// Yuv.crop (tools/src/yuv.ml:42-62) of one plane: dst[col, row] = src[clamp (col + x_pos), clamp (row + y_pos)]
__global__ __launch_bounds__(256) void k_crop(PlaneOp P) {
    const unsigned groups = (unsigned)(P.dw + 7) >> 3;
    unsigned wplane, wpiece;
    hvc::xcd_work(P.xcd_map, P.xcd_magic, wplane, wpiece);
    const unsigned t = wpiece * 256u + threadIdx.x;
    if (t >= groups * (unsigned)P.dh) return;
    const unsigned row = t / groups, g = t - row * groups;
    int sr = (int)row + P.y_pos;
    sr = sr < 0 ? 0 : sr >= P.sh ? P.sh - 1 : sr;
    const uint8_t *s = P.src + (size_t)wplane * P.src_ps + (size_t)sr * P.src_stride;
    uint8_t *d = P.dst + (size_t)wplane * P.dst_ps + (size_t)row * P.dst_stride + 8 * g;
    const int c0 = (int)(8 * g) + P.x_pos;
    if (P.vec && (int)(8 * g + 8) <= P.dw && c0 >= 0 && c0 + 8 <= P.sw) { // no column clamps: 8 bytes at once, from any alignment
        u2v w;
        __builtin_memcpy(&w, s + c0, 8);
        *reinterpret_cast<u2v *>(d) = w;
        return;
    }
    unsigned b[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        int c = (int)(8 * g) + i + P.x_pos;
        c = c < 0 ? 0 : c >= P.sw ? P.sw - 1 : c;
        b[i] = s[c];
    }
    if (P.vec && (int)(8 * g + 8) <= P.dw) {
        *reinterpret_cast<uint2 *>(d) = make_uint2(b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24), b[4] | (b[5] << 8) | (b[6] << 16) | (b[7] << 24));
        return;
    }
    for (int i = 0; i < 8 && (int)(8 * g) + i < P.dw; i++) d[i] = (uint8_t)b[i];
}

// Packed_422 (tools/src/packed_422.ml:6-44): a row of w luma samples is 2 w bytes, four per pair of pixels, with the luma
// samples at byte yo and yo + 2 and the chroma samples at uo and vo (yuy2 = 0 1 3, uyvy = 1 0 2, yvyu = 0 3 1).
struct PackedOp {
    const uint8_t *packed_in;
    uint8_t *packed_out;
    uint8_t *y, *u, *v;            // planar 4:2:2: w x h, (w / 2) x h, (w / 2) x h, tight
    const uint8_t *cy, *cu, *cv;   // the same planes as a source
    int w, h, yo, uo, vo;
    int vec;                       // bases and strides allow the lane's 16 / 8 / 4 / 4-byte pieces
    unsigned sel_y, sel_u, sel_v;  // v_perm_b32 selectors: two packed dwords -> 4 luma / 2 + 2 chroma samples
    unsigned sel_out[2];           //                        {luma dword, (u, v)} -> the packed dword of an even / odd pair
    int xcd_map;                   // hvc::xcd_work
    unsigned xcd_magic, pad;
    size_t packed_fs, planar_fs;   // bytes from frame to frame
    size_t luma_fs;                // ... of the luma plane, which may live in another array than the chroma planes
};
// One lane moves four pairs of pixels: 16 packed bytes <-> 8 luma + 4 + 4 chroma samples; a wave's pieces are contiguous
// runs of a row in all four arrays.  The byte shuffles are v_perm_b32 with selectors the host makes from (yo, uo, vo).
__global__ __launch_bounds__(256) void k_unpack422(PackedOp P) { // convert_to_planar :10-23
    const unsigned pairs = (unsigned)P.w >> 1, groups = (pairs + 3) >> 2;
    unsigned wframe, wpiece;
    hvc::xcd_work(P.xcd_map, P.xcd_magic, wframe, wpiece);
    const unsigned t = wpiece * 256u + threadIdx.x;
    if (t >= groups * (unsigned)P.h) return;
    const unsigned row = t / groups, g = t - row * groups;
    const uint8_t *s = P.packed_in + (size_t)wframe * P.packed_fs + (size_t)row * 2 * P.w + 16 * g;
    const size_t f = (size_t)wframe * P.planar_fs;
    uint8_t *y = P.y + (size_t)wframe * P.luma_fs + (size_t)row * P.w + 8 * g, *u = P.u + f + (size_t)row * pairs + 4 * g,
            *v = P.v + f + (size_t)row * pairs + 4 * g;
    if (P.vec && 4 * g + 4 <= pairs) {
        const u4v d = __builtin_nontemporal_load(reinterpret_cast<const u4v *>(s));
        const u2v yy = {__builtin_amdgcn_perm(d.y, d.x, P.sel_y), __builtin_amdgcn_perm(d.w, d.z, P.sel_y)};
        const unsigned u01 = __builtin_amdgcn_perm(d.y, d.x, P.sel_u), u23 = __builtin_amdgcn_perm(d.w, d.z, P.sel_u);
        const unsigned v01 = __builtin_amdgcn_perm(d.y, d.x, P.sel_v), v23 = __builtin_amdgcn_perm(d.w, d.z, P.sel_v);
        __builtin_nontemporal_store(yy, reinterpret_cast<u2v *>(y));
        __builtin_nontemporal_store(__builtin_amdgcn_perm(u23, u01, 0x05040100u), reinterpret_cast<unsigned *>(u));
        __builtin_nontemporal_store(__builtin_amdgcn_perm(v23, v01, 0x05040100u), reinterpret_cast<unsigned *>(v));
        return;
    }
    for (unsigned k = 0; k < 4 && 4 * g + k < pairs; k++) {
        y[2 * k] = s[4 * k + P.yo];
        y[2 * k + 1] = s[4 * k + P.yo + 2];
        u[k] = s[4 * k + P.uo];
        v[k] = s[4 * k + P.vo];
    }
}
__global__ __launch_bounds__(256) void k_pack422(PackedOp P) { // convert_from_planar :33-46
    const unsigned pairs = (unsigned)P.w >> 1, groups = (pairs + 3) >> 2;
    unsigned wframe, wpiece;
    hvc::xcd_work(P.xcd_map, P.xcd_magic, wframe, wpiece);
    const unsigned t = wpiece * 256u + threadIdx.x;
    if (t >= groups * (unsigned)P.h) return;
    const unsigned row = t / groups, g = t - row * groups;
    uint8_t *d = P.packed_out + (size_t)wframe * P.packed_fs + (size_t)row * 2 * P.w + 16 * g;
    const size_t f = (size_t)wframe * P.planar_fs;
    const uint8_t *y = P.cy + (size_t)wframe * P.luma_fs + (size_t)row * P.w + 8 * g, *u = P.cu + f + (size_t)row * pairs + 4 * g,
                  *v = P.cv + f + (size_t)row * pairs + 4 * g;
    if (P.vec && 4 * g + 4 <= pairs) {
        const u2v yy = __builtin_nontemporal_load(reinterpret_cast<const u2v *>(y));
        const unsigned uu = __builtin_nontemporal_load(reinterpret_cast<const unsigned *>(u));
        const unsigned vv = __builtin_nontemporal_load(reinterpret_cast<const unsigned *>(v));
        u4v o; // pair j: the lane's luma bytes 2 j, 2 j + 1 and its chroma bytes j
        o.x = __builtin_amdgcn_perm(yy.x, __builtin_amdgcn_perm(uu, vv, 0x0c0c0004u), P.sel_out[0]);
        o.y = __builtin_amdgcn_perm(yy.x, __builtin_amdgcn_perm(uu, vv, 0x0c0c0105u), P.sel_out[1]);
        o.z = __builtin_amdgcn_perm(yy.y, __builtin_amdgcn_perm(uu, vv, 0x0c0c0206u), P.sel_out[0]);
        o.w = __builtin_amdgcn_perm(yy.y, __builtin_amdgcn_perm(uu, vv, 0x0c0c0307u), P.sel_out[1]);
        __builtin_nontemporal_store(o, reinterpret_cast<u4v *>(d));
        return;
    }
    for (unsigned k = 0; k < 4 && 4 * g + k < pairs; k++) {
        d[4 * k + P.yo] = y[2 * k];
        d[4 * k + P.yo + 2] = y[2 * k + 1];
        d[4 * k + P.uo] = u[k];
        d[4 * k + P.vo] = v[k];
    }
}

// the launch of either: selectors, alignment, grid
template <class K>
hipError_t launch_packed_op(K kernel, PackedOp P, int n_frames, hipStream_t s) {
    const unsigned long long lanes = (unsigned long long)(((P.w >> 1) + 3) >> 2) * (unsigned long long)P.h;
    if (n_frames <= 0 || lanes == 0) return hipSuccess;
    const dim3 grid((unsigned)((lanes + 255) / 256), (unsigned)n_frames, 1);
    const unsigned yo = (unsigned)P.yo, uo = (unsigned)P.uo, vo = (unsigned)P.vo;
    P.sel_y = ((4 + yo + 2) << 24) | ((4 + yo) << 16) | ((yo + 2) << 8) | yo;
    P.sel_u = 0x0c0c0000u | ((4 + uo) << 8) | uo;
    P.sel_v = 0x0c0c0000u | ((4 + vo) << 8) | vo;
    for (unsigned h = 0; h < 2; h++) // S0 = the luma dword (selector bytes 4 .. 7), S1 = (u, v) in bytes 0, 1
        P.sel_out[h] = ((4 + 2 * h) << (8 * yo)) | ((4 + 2 * h + 1) << (8 * (yo + 2))) | (0u << (8 * uo)) | (1u << (8 * vo));
    const uintptr_t pk = (uintptr_t)(P.packed_in ? P.packed_in : P.packed_out), py = (uintptr_t)(P.cy ? P.cy : P.y),
                    pu = (uintptr_t)(P.cu ? P.cu : P.u), pv = (uintptr_t)(P.cv ? P.cv : P.v);
    P.vec = P.w % 8 == 0 && pk % 16 == 0 && py % 8 == 0 && pu % 4 == 0 && pv % 4 == 0 && P.packed_fs % 16 == 0 && P.planar_fs % 4 == 0 &&
            P.luma_fs % 8 == 0;
    P.xcd_map = hvc::xcd_map_for(grid.x, grid.y, P.xcd_magic);
    hipLaunchKernelGGL(kernel, grid, dim3(256), 0, s, P);
    return hipGetLastError();
}

// ---- two compositions Oconv makes through the 4:4:4 frame, without the frame (same size, no offset): what the full-size chroma
// plane would have held is formed in registers.  avg4 on four samples at once: avg4(a, b, c, d) = v_lerp_u8(avg2(a, b), (c + d)
// >> 1, r) with r = ~(a ^ b) | (c ^ d) in bit 0 of every byte (tests/test_guard_bounds.py::test_avg4_by_lerp_identity).
__device__ __forceinline__ unsigned avg4x4(unsigned a, unsigned b, unsigned c, unsigned d) {
    return __builtin_amdgcn_lerp(__builtin_amdgcn_lerp(a, b, 0x01010101u), __builtin_amdgcn_lerp(c, d, 0u), ~(a ^ b) | (c ^ d));
}
// every sample's right neighbour in the lane's two dwords; `next` = the sample behind them (the last column: itself)
__device__ __forceinline__ void right_of(unsigned a0, unsigned a1, unsigned next, unsigned &b0, unsigned &b1) {
    b0 = __builtin_amdgcn_alignbyte(a1, a0, 1);
    b1 = (a1 >> 8) | (next << 24);
}
// 4:2:0 -> 4:2:2 chroma: subsample_h2 (supersample_hv2 src) (planar_444.ml:18-23 of :82-103): dst = sw x 2 sh
//   dst[c, 2r]     = avg2 a (avg2 a b)                         a = src[c, r], b = src[c + 1, r], c' = src[c, r + 1],
//   dst[c, 2r + 1] = avg2 (avg2 a c') (avg4 a b c' d)          d = src[c + 1, r + 1]; last column / row: the sample itself
__global__ __launch_bounds__(256) void k_chroma_420_to_422(PlaneOp P) {
    const unsigned groups = (unsigned)(P.sw + 7) >> 3;
    unsigned wplane, wpiece;
    hvc::xcd_work(P.xcd_map, P.xcd_magic, wplane, wpiece);
    const unsigned t = wpiece * 256u + threadIdx.x;
    if (t >= groups * (unsigned)P.sh) return;
    const unsigned row = t / groups, g = t - row * groups, row2 = min(row + 1u, (unsigned)P.sh - 1u);
    const uint8_t *s1 = P.src + (size_t)wplane * P.src_ps + (size_t)row * P.src_stride + 8 * g;
    const uint8_t *s2 = P.src + (size_t)wplane * P.src_ps + (size_t)row2 * P.src_stride + 8 * g;
    uint8_t *d1 = P.dst + (size_t)wplane * P.dst_ps + (size_t)(2 * row) * P.dst_stride + 8 * g, *d2 = d1 + P.dst_stride;
    if (P.vec && (int)(8 * g + 8) <= P.sw) {
        const u2v a = *reinterpret_cast<const u2v *>(s1), cc = *reinterpret_cast<const u2v *>(s2);
        const bool last = (int)(8 * g + 8) == P.sw;
        unsigned b0, b1, e0, e1;
        right_of(a.x, a.y, last ? a.y >> 24 : (unsigned)s1[8], b0, b1);
        right_of(cc.x, cc.y, last ? cc.y >> 24 : (unsigned)s2[8], e0, e1);
        const u2v o1 = {avg2x4(a.x, avg2x4(a.x, b0)), avg2x4(a.y, avg2x4(a.y, b1))};
        const u2v o2 = {avg2x4(avg2x4(a.x, cc.x), avg4x4(a.x, b0, cc.x, e0)), avg2x4(avg2x4(a.y, cc.y), avg4x4(a.y, b1, cc.y, e1))};
        __builtin_nontemporal_store(o1, reinterpret_cast<u2v *>(d1));
        __builtin_nontemporal_store(o2, reinterpret_cast<u2v *>(d2));
        return;
    }
    for (int i = 0; i < 8 && (int)(8 * g) + i < P.sw; i++) {
        const int c = (int)(8 * g) + i, cn = c + 1 < P.sw ? c + 1 : P.sw - 1;
        const uint8_t *r1 = s1 - 8 * g, *r2 = s2 - 8 * g;
        const unsigned a = r1[c], b = r1[cn], cv = r2[c], dv = r2[cn];
        d1[i] = (uint8_t)((a + ((a + b + 1u) >> 1) + 1u) >> 1);
        d2[i] = (uint8_t)((((a + cv + 1u) >> 1) + ((a + b + cv + dv + 2u) >> 2) + 1u) >> 1);
    }
}
// 4:2:2 -> 4:2:0 chroma: subsample_hv2 (supersample_h2 src) (planar_444.ml:69-80 of :25-33): dst = sw x (sh / 2)
//   dst[c, r] = avg4 a (avg2 a b) a' (avg2 a' b')    a, b = src[c, 2r], src[c + 1, 2r]; a', b' the same of row 2r + 1
__global__ __launch_bounds__(256) void k_chroma_422_to_420(PlaneOp P) {
    const unsigned groups = (unsigned)(P.sw + 7) >> 3;
    unsigned wplane, wpiece;
    hvc::xcd_work(P.xcd_map, P.xcd_magic, wplane, wpiece);
    const unsigned t = wpiece * 256u + threadIdx.x;
    if (t >= groups * (unsigned)P.dh) return;
    const unsigned row = t / groups, g = t - row * groups;
    const uint8_t *s1 = P.src + (size_t)wplane * P.src_ps + (size_t)(2 * row) * P.src_stride + 8 * g, *s2 = s1 + P.src_stride;
    uint8_t *d = P.dst + (size_t)wplane * P.dst_ps + (size_t)row * P.dst_stride + 8 * g;
    if (P.vec && (int)(8 * g + 8) <= P.sw) {
        const u2v a = __builtin_nontemporal_load(reinterpret_cast<const u2v *>(s1)), cc = __builtin_nontemporal_load(reinterpret_cast<const u2v *>(s2));
        const bool last = (int)(8 * g + 8) == P.sw;
        unsigned b0, b1, e0, e1;
        right_of(a.x, a.y, last ? a.y >> 24 : (unsigned)s1[8], b0, b1);
        right_of(cc.x, cc.y, last ? cc.y >> 24 : (unsigned)s2[8], e0, e1);
        const u2v o = {avg4x4(a.x, avg2x4(a.x, b0), cc.x, avg2x4(cc.x, e0)), avg4x4(a.y, avg2x4(a.y, b1), cc.y, avg2x4(cc.y, e1))};
        __builtin_nontemporal_store(o, reinterpret_cast<u2v *>(d));
        return;
    }
    for (int i = 0; i < 8 && (int)(8 * g) + i < P.sw; i++) {
        const int c = (int)(8 * g) + i, cn = c + 1 < P.sw ? c + 1 : P.sw - 1;
        const uint8_t *r1 = s1 - 8 * g, *r2 = s2 - 8 * g;
        const unsigned a = r1[c], b = r1[cn], a2 = r2[c], b2 = r2[cn];
        d[i] = (uint8_t)((a + ((a + b + 1u) >> 1) + a2 + ((a2 + b2 + 1u) >> 1) + 2u) >> 2);
    }
}

enum OpKind { OP_SUB420, OP_SUB422, OP_UP422, OP_CROP, OP_C420_422, OP_C422_420 };

// the lanes of one launch: groups of 8 samples per row
hipError_t launch_plane_op(OpKind kind, PlaneOp P, int n_planes, hipStream_t s) {
    const bool by_source = kind == OP_UP422 || kind == OP_C420_422; // (a lane = 8 source samples; else 8 destination samples)
    const int cols = by_source ? P.sw : P.dw, rows = by_source ? P.sh : P.dh;
    if (n_planes <= 0 || cols <= 0 || rows <= 0) return hipSuccess;
    const unsigned long long lanes = (unsigned long long)((cols + 7) >> 3) * (unsigned long long)rows;
    const dim3 grid((unsigned)((lanes + 255) / 256), (unsigned)n_planes, 1);
    const bool same_w = kind == OP_C420_422 || kind == OP_C422_420; // (8 bytes in, 8 out)
    const size_t sa = kind == OP_UP422 || same_w ? 8 : 16, da = kind == OP_UP422 ? 16 : 8; // bytes a lane loads / stores at once
    P.vec = kind == OP_CROP ? ((uintptr_t)P.dst % 8 == 0 && P.dst_stride % 8 == 0 && P.dst_ps % 8 == 0)
                            : ((uintptr_t)P.src % sa == 0 && P.src_stride % sa == 0 && P.src_ps % sa == 0 &&
                               (uintptr_t)P.dst % da == 0 && P.dst_stride % da == 0 && P.dst_ps % da == 0);
    P.xcd_map = hvc::xcd_map_for(grid.x, grid.y, P.xcd_magic);
    switch (kind) {
    case OP_SUB420: hipLaunchKernelGGL(k_subsample420, grid, dim3(256), 0, s, P); break;
    case OP_SUB422: hipLaunchKernelGGL(k_subsample422, grid, dim3(256), 0, s, P); break;
    case OP_UP422: hipLaunchKernelGGL(k_upsample422, grid, dim3(256), 0, s, P); break;
    case OP_CROP: hipLaunchKernelGGL(k_crop, grid, dim3(256), 0, s, P); break;
    case OP_C420_422: hipLaunchKernelGGL(k_chroma_420_to_422, grid, dim3(256), 0, s, P); break;
    case OP_C422_420: hipLaunchKernelGGL(k_chroma_422_to_420, grid, dim3(256), 0, s, P); break;
    }
    return hipGetLastError();
}

// One plane operation behind the C ABI: device memory as it is, host memory through the context's staging buffers.
int plane_op(hvc_ctx *c, OpKind kind, const uint8_t *src, int sw, int sh, size_t src_stride, int x_pos, int y_pos, uint8_t *dst,
             int dw, int dh, size_t dst_stride, int n_planes, size_t src_ps, size_t dst_ps, int where) {
    if (!c || !src || !dst || sw < 1 || sh < 1 || n_planes < 0) return HVC_E_INVALID_ARG;
    if (where != HVC_MEM_HOST && where != HVC_MEM_DEVICE) return HVC_E_INVALID_ARG;
    if (src_stride < (size_t)sw || dst_stride < (size_t)dw) return HVC_E_INVALID_ARG;
    if (n_planes == 0 || dw < 1 || dh < 1) return HVC_OK; // (a plane of one column has no half: nothing to write)
    if (n_planes > 65535) return HVC_E_TOO_LARGE;
    if ((unsigned long long)((dw > sw ? dw : sw) + 7) / 8 * (unsigned long long)(dh > sh ? dh : sh) >= (1ull << 31)) return HVC_E_TOO_LARGE;
    if (!src_ps) src_ps = src_stride * (size_t)sh;
    if (!dst_ps) dst_ps = dst_stride * (size_t)dh;
    DeviceGuard g(c->device);
    if (!g.ok) return fail_hip(c, hipErrorInvalidDevice);
    PlaneOp P;
    std::memset(&P, 0, sizeof P);
    P.sw = sw, P.sh = sh, P.dw = dw, P.dh = dh, P.x_pos = x_pos, P.y_pos = y_pos;
    P.src_stride = src_stride, P.dst_stride = dst_stride, P.src_ps = src_ps, P.dst_ps = dst_ps;
    if (where == HVC_MEM_DEVICE) {
        P.src = src;
        P.dst = dst;
        HIPCHK(c, launch_plane_op(kind, P, n_planes, c->stream));
        return HVC_OK;
    }
    const size_t sbytes = (size_t)(n_planes - 1) * src_ps + (size_t)(sh - 1) * src_stride + (size_t)sw;
    const size_t dbytes = (size_t)(n_planes - 1) * dst_ps + (size_t)(dh - 1) * dst_stride + (size_t)dw;
    int r = grow(c, &c->d_in, &c->in_cap, sbytes);
    if (r) return r;
    if ((r = grow(c, &c->d_out, &c->out_cap, dbytes))) return r;
    HIPCHK(c, hipMemcpyAsync(c->d_in, src, sbytes, hipMemcpyHostToDevice, c->stream));
    P.src = (const uint8_t *)c->d_in;
    P.dst = (uint8_t *)c->d_out;
    HIPCHK(c, launch_plane_op(kind, P, n_planes, c->stream));
    for (int p = 0; p < n_planes; p++) // (only what was written: the caller's padding stays)
        HIPCHK(c, hipMemcpy2DAsync(dst + (size_t)p * dst_ps, dst_stride, (uint8_t *)c->d_out + (size_t)p * dst_ps, dst_stride,
                                   (size_t)dw, (size_t)dh, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return HVC_OK;
}

// Yuv_format (tools/src/yuv_format.ml): planar C420 / C422 / C444, packed YUY2 / UYVY / YVYU
bool is_packed(int f) { return f == HVC_YUV_YUY2 || f == HVC_YUV_UYVY || f == HVC_YUV_YVYU; }
bool is_format(int f) { return f == HVC_YUV_420 || f == HVC_YUV_422 || f == HVC_YUV_444 || is_packed(f); }
void chroma_size(int f, int w, int h, int &cw, int &ch) { // Planar.create :37-53 (integer halves); packed = 4:2:2
    cw = f == HVC_YUV_444 ? w : w / 2;
    ch = f == HVC_YUV_420 ? h / 2 : h;
}
size_t frame_bytes(int f, int w, int h) {
    if (is_packed(f)) return (size_t)2 * w * h; // Packed.create: a plane of 2 w x h
    int cw, ch;
    chroma_size(f, w, h, cw, ch);
    return (size_t)w * h + 2 * (size_t)cw * ch;
}
// what Oconv.input / output raise on: Yuv.assert_is_420 / _422 (tools/src/yuv.ml:90-116) want wy = 2 wu (and hy = 2 hu)
bool size_fits(int f, int w, int h) {
    if (w < 1 || h < 1) return false;
    if (f == HVC_YUV_444) return true;
    if (w & 1) return false;
    return f != HVC_YUV_420 || !(h & 1);
}
void packed_offsets(int f, int &yo, int &uo, int &vo) { // packed_422.ml:6-8
    if (f == HVC_YUV_YUY2) yo = 0, uo = 1, vo = 3;
    else if (f == HVC_YUV_UYVY) yo = 1, uo = 0, vo = 2;
    else yo = 0, uo = 3, vo = 1;
}

struct Planes { // three planes of a batch of frames: pointer to frame 0's, samples per row = the plane's width, bytes from frame to frame
    const uint8_t *p[3];
    int w[3], h[3];
    size_t fs[3];
};

} // namespace

extern "C" {

int hvc_subsample420(hvc_ctx *c, const uint8_t *src, int sw, int sh, size_t src_stride, uint8_t *dst, size_t dst_stride, int n_planes,
                     size_t src_ps, size_t dst_ps, int where) try {
    return plane_op(c, OP_SUB420, src, sw, sh, src_stride, 0, 0, dst, sw / 2, sh / 2, dst_stride, n_planes, src_ps, dst_ps, where);
} HVC_ABI_CATCH

int hvc_subsample422(hvc_ctx *c, const uint8_t *src, int sw, int sh, size_t src_stride, uint8_t *dst, size_t dst_stride, int n_planes,
                     size_t src_ps, size_t dst_ps, int where) try {
    return plane_op(c, OP_SUB422, src, sw, sh, src_stride, 0, 0, dst, sw / 2, sh, dst_stride, n_planes, src_ps, dst_ps, where);
} HVC_ABI_CATCH

int hvc_upsample422(hvc_ctx *c, const uint8_t *src, int cw, int h, size_t src_stride, uint8_t *dst, size_t dst_stride, int n_planes,
                    size_t src_ps, size_t dst_ps, int where) try {
    if (cw > (1 << 29)) return HVC_E_TOO_LARGE;
    return plane_op(c, OP_UP422, src, cw, h, src_stride, 0, 0, dst, 2 * cw, h, dst_stride, n_planes, src_ps, dst_ps, where);
} HVC_ABI_CATCH

int hvc_crop_planes(hvc_ctx *c, const uint8_t *src, int sw, int sh, size_t src_stride, int x_pos, int y_pos, uint8_t *dst, int dw, int dh,
                    size_t dst_stride, int n_planes, size_t src_ps, size_t dst_ps, int where) try {
    if (dw < 0 || dh < 0) return HVC_E_INVALID_ARG;
    return plane_op(c, OP_CROP, src, sw, sh, src_stride, x_pos, y_pos, dst, dw, dh, dst_stride, n_planes, src_ps, dst_ps, where);
} HVC_ABI_CATCH

int hvc_yuv_frame_bytes(int format, int width, int height, size_t *bytes) try {
    if (!bytes || !is_format(format) || width < 0 || height < 0) return HVC_E_INVALID_ARG;
    *bytes = frame_bytes(format, width, height);
    return HVC_OK;
} HVC_ABI_CATCH

// Oconv.main's loop body for n_frames frames at once (oconv.ml:111-133): input -> 4:4:4 -> Yuv.crop -> output.
int hvc_yuv_convert(hvc_ctx *c, const uint8_t *src, int src_format, int src_w, int src_h, int x_off, int y_off, uint8_t *dst,
                    int dst_format, int dst_w, int dst_h, int n_frames, int where) try {
    if (!c || !src || !dst || n_frames < 0 || !is_format(src_format) || !is_format(dst_format)) return HVC_E_INVALID_ARG;
    if (where != HVC_MEM_HOST && where != HVC_MEM_DEVICE) return HVC_E_INVALID_ARG;
    if (!size_fits(src_format, src_w, src_h) || !size_fits(dst_format, dst_w, dst_h)) return HVC_E_INVALID_ARG;
    if (src_w > 65535 || src_h > 65535 || dst_w > 65535 || dst_h > 65535 || n_frames > 65535) return HVC_E_TOO_LARGE;
    if (n_frames == 0) return HVC_OK;
    DeviceGuard g(c->device);
    if (!g.ok) return fail_hip(c, hipErrorInvalidDevice);
    const size_t in_fs = frame_bytes(src_format, src_w, src_h), out_fs = frame_bytes(dst_format, dst_w, dst_h);
    const size_t sp = (size_t)src_w * src_h, dp = (size_t)dst_w * dst_h; // samples of a full-size plane, in and out
    const uint8_t *d_src = src;
    uint8_t *d_dst = dst;
    int r;
    if (where == HVC_MEM_HOST) {
        if ((r = grow(c, &c->d_in, &c->in_cap, in_fs * (size_t)n_frames))) return r;
        if ((r = grow(c, &c->d_out, &c->out_cap, out_fs * (size_t)n_frames))) return r;
        HIPCHK(c, hipMemcpyAsync(c->d_in, src, in_fs * (size_t)n_frames, hipMemcpyHostToDevice, c->stream));
        d_src = (const uint8_t *)c->d_in;
        d_dst = (uint8_t *)c->d_out;
    }
    // scratch: [planar 4:2:2 of a packed input: 2 sp][full-size chroma of the input: 2 sp] and
    //          [cropped 4:4:4 chroma, or all three planes for a packed output: 3 dp][planar 4:2:2 of a packed output: 2 dp]
    const size_t a_fs = 4 * sp, b_fs = 5 * dp;
    if ((r = grow(c, &c->d_aux, &c->aux_cap, a_fs * (size_t)n_frames))) return r;
    if ((r = grow(c, &c->d_aux2, &c->aux2_cap, b_fs * (size_t)n_frames))) return r;
    uint8_t *const A = (uint8_t *)c->d_aux, *const B = (uint8_t *)c->d_aux2;
    hipStream_t st = c->stream;
    auto op = [&](OpKind kind, const uint8_t *s, int sw, int sh, size_t s_fs, uint8_t *d, int dw, int dh, size_t d_fs,
                  size_t s_stride = 0) -> hipError_t { // (s_stride: the source is a window of wider rows)
        PlaneOp P;
        std::memset(&P, 0, sizeof P);
        P.src = s, P.dst = d, P.sw = sw, P.sh = sh, P.dw = dw, P.dh = dh, P.x_pos = x_off, P.y_pos = y_off;
        P.src_stride = s_stride ? s_stride : (size_t)sw, P.dst_stride = (size_t)dw, P.src_ps = s_fs, P.dst_ps = d_fs;
        return launch_plane_op(kind, P, n_frames, st);
    };
    auto copy_plane = [&](const uint8_t *s, size_t s_fs, uint8_t *d, size_t d_fs, size_t bytes) -> hipError_t { // Plane.blit per frame
        return hipMemcpy2DAsync(d, d_fs, s, s_fs, bytes, (size_t)n_frames, hipMemcpyDeviceToDevice, st);
    };

    // where the output's planes live (known first: an input stage whose result IS the output writes it there)
    const bool same = dst_w == src_w && dst_h == src_h && x_off == 0 && y_off == 0; // the crop is the identity
    int dcw, dch;
    chroma_size(dst_format, dst_w, dst_h, dcw, dch);
    const bool packed_out = is_packed(dst_format);
    uint8_t *const oy = packed_out ? B + 3 * dp : d_dst; // the output's planar luma plane
    const size_t o_fs = packed_out ? b_fs : out_fs;
    uint8_t *const ou = oy + dp, *const ov = ou + (size_t)dcw * dch;
    const int out_planar = packed_out ? HVC_YUV_422 : dst_format;
    const int in_planar = is_packed(src_format) ? HVC_YUV_422 : src_format;
    const bool direct = same && out_planar == HVC_YUV_444 && in_planar != HVC_YUV_444; // the input stage's full-size chroma planes ARE the output's
    const bool luma_direct = same && is_packed(src_format) && !packed_out; // ... and the unpacked luma plane is
    // 4:2:0 <-> 4:2:2 at the same size: the chroma planes go from one sampling to the other in one kernel, the full-size
    // planes in between formed in registers (k_chroma_420_to_422 / _422_to_420)
    const bool chroma_fused = same && ((in_planar == HVC_YUV_420 && out_planar == HVC_YUV_422) || (in_planar == HVC_YUV_422 && out_planar == HVC_YUV_420));

    // ---- Oconv.input (oconv.ml:12-28): the frame as three full-size planes
    Planes in;
    int scw, sch;
    chroma_size(src_format, src_w, src_h, scw, sch);
    const uint8_t *py = d_src, *pu = d_src + sp, *pv = d_src + sp + (size_t)scw * sch; // planar input
    size_t p_fs = in_fs;
    int planar = src_format;
    if (is_packed(src_format)) { // Packed_422.convert_to_planar into the scratch's first part
        PackedOp K;
        std::memset(&K, 0, sizeof K);
        K.packed_in = d_src, K.y = luma_direct ? oy : A, K.u = A + sp, K.v = A + sp + sp / 2, K.w = src_w, K.h = src_h;
        packed_offsets(src_format, K.yo, K.uo, K.vo);
        K.packed_fs = in_fs, K.planar_fs = a_fs, K.luma_fs = luma_direct ? o_fs : a_fs;
        HIPCHK(c, launch_packed_op(k_unpack422, K, n_frames, st));
        py = A, pu = A + sp, pv = A + sp + sp / 2, p_fs = a_fs, planar = HVC_YUV_422;
    }
    in.p[0] = py, in.w[0] = src_w, in.h[0] = src_h, in.fs[0] = p_fs;
    for (int k = 1; k < 3; k++) in.w[k] = src_w, in.h[k] = src_h;
    if (planar == HVC_YUV_444 || chroma_fused) { // (fused: the planes as they are, at their own sampling)
        in.p[1] = pu, in.p[2] = pv, in.fs[1] = in.fs[2] = p_fs;
    } else {
        uint8_t *const up[2] = {direct ? ou : A + 2 * sp, direct ? ov : A + 3 * sp};
        const size_t up_fs = direct ? o_fs : a_fs;
        const uint8_t *const cp[2] = {pu, pv};
        for (int k = 0; k < 2; k++) {
            if (planar == HVC_YUV_420) { // Planar_444.convert_from_420 :122-131 (K2)
                hvc::UpsampleParams U;
                std::memset(&U, 0, sizeof U);
                U.src = cp[k], U.dst = up[k], U.cw = src_w / 2, U.ch = src_h / 2, U.n_planes = n_frames;
                U.src_stride = (size_t)(src_w / 2), U.dst_stride = (size_t)src_w, U.src_ps = p_fs, U.dst_ps = up_fs;
                HIPCHK(c, hvc::launch_upsample420(U, st));
            } else { // convert_from_422 :55-67
                HIPCHK(c, op(OP_UP422, cp[k], src_w / 2, src_h, p_fs, up[k], src_w, src_h, up_fs));
            }
            in.p[k + 1] = up[k], in.fs[k + 1] = up_fs;
        }
    }

    // ---- Yuv.crop (yuv.ml:42-62) and Oconv.output (oconv.ml:38-51)
    // luma: straight to its place
    const bool luma_in_place = same && packed_out; // the packing kernel reads the input's luma plane where it is
    if (luma_direct || luma_in_place) {
    } else if (same) HIPCHK(c, copy_plane(in.p[0], in.fs[0], oy, o_fs, dp));
    else HIPCHK(c, op(OP_CROP, in.p[0], src_w, src_h, in.fs[0], oy, dst_w, dst_h, o_fs));
    // chroma: full size after the crop, then the output's sampling.  A crop window that lies inside the source (nothing to
    // clamp) and keeps the 16-byte pieces of the sub-sampling kernels aligned is not materialised: the kernels read the window
    // in place, with the source's row stride.
    const bool window = !same && out_planar != HVC_YUV_444 && x_off >= 0 && y_off >= 0 && x_off + dst_w <= src_w &&
                        y_off + dst_h <= src_h && x_off % 16 == 0 && src_w % 16 == 0;
    for (int k = 1; k < 3 && chroma_fused; k++) {
        uint8_t *const od = k == 1 ? ou : ov;
        if (in_planar == HVC_YUV_420) HIPCHK(c, op(OP_C420_422, in.p[k], src_w / 2, src_h / 2, in.fs[k], od, dst_w / 2, dst_h, o_fs));
        else HIPCHK(c, op(OP_C422_420, in.p[k], src_w / 2, src_h, in.fs[k], od, dst_w / 2, dst_h / 2, o_fs));
    }
    for (int k = 1; k < 3 && !direct && !chroma_fused; k++) {
        uint8_t *const od = k == 1 ? ou : ov;
        const uint8_t *full = in.p[k];
        size_t full_fs = in.fs[k], full_stride = 0;
        if (window) {
            full += (size_t)y_off * src_w + x_off, full_stride = (size_t)src_w;
        } else if (!same) {
            uint8_t *const cropped = out_planar == HVC_YUV_444 ? od : B + (size_t)(k - 1) * dp;
            HIPCHK(c, op(OP_CROP, in.p[k], src_w, src_h, in.fs[k], cropped, dst_w, dst_h, out_planar == HVC_YUV_444 ? o_fs : b_fs));
            full = cropped, full_fs = out_planar == HVC_YUV_444 ? o_fs : b_fs;
        }
        if (out_planar == HVC_YUV_444) {
            if (same) HIPCHK(c, copy_plane(full, full_fs, od, o_fs, dp));
        } else if (out_planar == HVC_YUV_420) { // convert_to_420 :105-116
            HIPCHK(c, op(OP_SUB420, full, dst_w, dst_h, full_fs, od, dst_w / 2, dst_h / 2, o_fs, full_stride));
        } else { // convert_to_422 :35-44
            HIPCHK(c, op(OP_SUB422, full, dst_w, dst_h, full_fs, od, dst_w / 2, dst_h, o_fs, full_stride));
        }
    }
    if (packed_out) { // Packed_422.convert_from_planar
        PackedOp K;
        std::memset(&K, 0, sizeof K);
        K.packed_out = d_dst, K.cy = luma_in_place ? in.p[0] : oy, K.cu = ou, K.cv = ov, K.w = dst_w, K.h = dst_h;
        packed_offsets(dst_format, K.yo, K.uo, K.vo);
        K.packed_fs = out_fs, K.planar_fs = b_fs, K.luma_fs = luma_in_place ? in.fs[0] : b_fs;
        HIPCHK(c, launch_packed_op(k_pack422, K, n_frames, st));
    }
    if (where == HVC_MEM_HOST) {
        HIPCHK(c, hipMemcpyAsync(dst, d_dst, out_fs * (size_t)n_frames, hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
    }
    return HVC_OK;
} HVC_ABI_CATCH

} // extern "C"
