// hvc_kernels.h -- kernel parameter blocks and launcher prototypes (internal).
#ifndef HVC_KERNELS_H
#define HVC_KERNELS_H

#include <hip/hip_runtime.h>
#include <cstdlib>
#include <stddef.h>
#include <stdint.h>

#define HVC_MAX_COMP 4
#define HVC_MAX_QTABS 4
#define HVC_GUARD_D ((1 << 17) - 1) /* fast kernel: largest |dequantised coefficient| it is proved for */
#define HVC_TILE 256 /* blocks per workgroup = threads per workgroup (one block per lane) */

#include "hvc_idct_spec.h" /* HVC_GUARD_D_PACKED: packed kernel, |dequantised coefficient| must fit int16; the pair order */

namespace hvc {

// jpeg/model/src/zigzag.ml:71-137  forward[raster] = zz  (host-side copy for table preparation)
static const unsigned char HVC_ZF[64] = {
    0,  1,  5,  6,  14, 15, 27, 28, 2,  4,  7,  13, 16, 26, 29, 42, 3,  8,  12, 17, 25, 30,
    41, 43, 9,  11, 18, 24, 31, 40, 44, 53, 10, 19, 23, 32, 39, 45, 52, 54, 20, 22, 33, 38,
    46, 51, 55, 60, 21, 34, 37, 47, 50, 56, 59, 61, 35, 36, 48, 49, 57, 58, 62, 63};

// One component plane of a frame as the kernels see it (Decoder.Component.t,
// jpeg/model/src/decoder.ml:167-187, reduced to geometry).
struct CompK {
    int bw, bh;        // plane size in 8x8 blocks
    int nblk;          // bw * bh
    int tile0;         // first tile (of HVC_TILE blocks) of this component inside a frame
    unsigned magic;    // ceil(2^32 / bw): by = umulhi(b, magic) for b < nblk (bw*bw*bh < 2^32)
    int qtab;          // table index
    size_t coef_off;   // int16 elements from the frame's coefficient record
    size_t plane_off;  // bytes from the frame's pixel record
    size_t stride;     // bytes per pixel row
};

struct DecodeParams {
    const int16_t *coefs;
    uint8_t *pixels;
    size_t coef_fs;   // int16 elements between frames
    size_t pixel_fs;  // bytes between frames
    int n_frames, n_comp, tiles_per_frame;
    int kernel_sel;   // 0 = k_decode_packed (default), 1 = k_decode_fast (hvc_set_decode_kernel; host-side only)
    CompK comp[HVC_MAX_COMP];
    int qt[HVC_MAX_QTABS * 64];  // quantiser tables, zig-zag order (kernarg segment -> scalar loads)
    int ethr[HVC_MAX_QTABS];     // per table: largest coefficient energy k_decode_fast accepts
    // k_decode_packed: per table and row r, the quantiser entries of the row's operand pairs
    // (raster 8r+1, 8r+7) (8r+5, 8r+3) (8r+2, 8r+6) (8r+0, 8r+4) as lo | hi << 16
    unsigned qpair[HVC_MAX_QTABS * 32];
    int ethr_packed[HVC_MAX_QTABS];  // largest coefficient energy k_decode_packed accepts
    unsigned *fix_count;       // device: number of entries in fix_list (this call's counter)
    unsigned *fix_count_next;  // device: the next call's counter, cleared by this call's wide kernel
    unsigned *fix_list;   // device: global block ids needing the wide kernel
    // Optional (batch pipeline behind the GPU Huffman reader): the blocks' absolute DC values in a compact array,
    // dc_plane[frame * dc_fs + (block's coefficient offset inside the frame record) / 64], read INSTEAD of coefficient
    // 0 of the record (which then holds the DC difference as the reader's write pass left it): the reader's DC pass
    // writes 2 bytes per block into a compact array instead of 2 bytes into every 128-byte record.
    const int16_t *dc_plane;
    size_t dc_fs;
    // hvc_last_wide_blocks: the fix-up kernels of ONE call add their list lengths up here (a call may be cut into several
    // launches, each with its own counter): the first launch of a call stores, the others add (stream order, no memset)
    unsigned long long *wide_total;
    int wide_first;
    int xcd_map;      // > 0: workgroups -> (frame, tile) by xcd_work(): every XCD takes runs of 2^(xcd_map - 1) tiles; 0: as dispatched
    unsigned xcd_magic; // ceil(2^32 / tiles per frame)
};

struct EncodeParams {
    const uint8_t *pixels;
    int16_t *coefs;
    size_t coef_fs, pixel_fs;
    int n_frames, n_comp, tiles_per_frame, xcd_map;
    unsigned xcd_magic, pad;
    CompK comp[HVC_MAX_COMP];
    float qrcp[HVC_MAX_QTABS * 64];   // fl((1 + 2^-16) / (4*q)), zig-zag order (kernarg segment)
};

struct UpsampleParams {
    const uint8_t *src;
    uint8_t *dst;
    int cw, ch, n_planes, xcd_map;
    size_t src_stride, dst_stride, src_ps, dst_ps;
    unsigned xcd_magic, pad;
};

// One plane of a 4:2:0 coefficient record decoded straight to a tight 4:4:4 frame
// (Decoder.get_yuv_frame's crop, decoder.ml:403-420, and Planar_444.convert_from_420,
// tools/src/planar_444.ml:82-131, folded into the block stage).
struct Plane444K {
    int bw;           // blocks per row of the coefficient plane (padded geometry, decoder.ml:304-345)
    int cbw, cbh;     // block columns / rows that intersect the crop
    int aw, ah;       // cropped plane size in samples (luma W x H, chroma W/2 x H/2)
    int qtab;
    size_t coef_off;  // int16 elements from the frame's coefficient record
    size_t out_off;   // bytes from the frame's output record (plane p sits at p * W * H)
};

/* chroma workgroup tile of the fused kernel: (64 * nw) x 4 blocks, nw = 1, 2 or 4 waves side by side per block row
 * (workgroup = 256 * nw lanes); horizontally consecutive tiles overlap by one block column */
#define HVC_444_TILE_BW 64
#define HVC_444_TILE_BH 4

struct Decode444Params {
    const int16_t *coefs;
    uint8_t *out;
    size_t coef_fs;   // int16 elements between frames
    size_t out_fs;    // bytes between output frames
    int n_frames, tiles_per_frame;
    int width, height;     // luma crop = size of all three output planes
    int y_tiles;           // luma: linear tiles of HVC_TILE blocks over cbw * cbh blocks
    unsigned y_magic;      // ceil(2^32 / cbw)
    int c_tiles_x, c_tiles_y; // chroma: tiles of 64 x 4 blocks (x step 63) over cbw x cbh, per plane
    unsigned c_magic;      // ceil(2^32 / c_tiles_x)
    int skip;              // measurements only (HVC_444_ONLY): 1 = luma tiles return at once, 2 = chroma tiles do; 0 = the kernel
    int nw;                // workgroup = 256 * nw lanes: luma tiles of that many blocks, chroma tiles (64 * nw) x 4 blocks.
                           // Chosen so that ONE chroma tile spans the output row where it can (1080p: nw = 2, 4K: nw = 4):
                           // a workgroup then writes whole rows, measured 58 -> 66 % of the HBM peak for the chroma half
    int tile0;             // added to blockIdx.x: 0, or y_tiles when the luma tiles run in a kernel of their own
    Plane444K pl[3];
    int qt[HVC_MAX_QTABS * 64];
    unsigned qpair[HVC_MAX_QTABS * 32];
    int ethr_packed[HVC_MAX_QTABS];
    unsigned *fix_count, *fix_count_next, *fix_list;
    const int16_t *dc_plane; // as in DecodeParams
    size_t dc_fs;
    unsigned long long *wide_total; // as in DecodeParams
    int wide_first;
    int xcd_map;                    // as in DecodeParams (other workgroup orders were measured and closed: profiles/r05c_fused_order.txt)
    unsigned xcd_magic;
};

// HVC_XCD_RUN in the environment (A/B switch): 0 = workgroups as dispatched; R = runs of R tiles per XCD (a power of two).
// Default: 16 for the block-per-lane kernels (K1, K3, K2), 64 for the fused 4:4:4 kernel, whose tiles are two or four times as
// large and mix 2:1 with 1:2 traffic (HVC_XCD_RUN_444 overrides that one) -- each the best of 16 / 32 / 64 in same-box A/Bs.
inline int xcd_map_run(bool fused = false) {
    static const int run = [] { const char *v = getenv("HVC_XCD_RUN"); return v ? atoi(v) : -1; }();
    static const int run444 = [] { const char *v = getenv("HVC_XCD_RUN_444"); return v ? atoi(v) : -1; }();
    if (fused) return run444 >= 0 ? run444 : run >= 0 ? run : 64;
    return run >= 0 ? run : 16;
}
// Workgroup -> (frame, tile).  The dispatcher deals a grid's workgroups round-robin over the 8 XCDs in linear order (x
// fastest, then y): taken as they come, consecutive tiles land behind eight different L2s and eight interleaved request
// streams sweep every stretch of memory.  Instead every XCD takes RUNS of R = 2^(map - 1) consecutive tiles of the batch's
// linear (frame, tile) order: workgroup id = y * tiles + x has XCD id % 8 and is that XCD's k-th (k = id / 8); it takes
// position k % R of the XCD's (k / R)-th run, and the runs of the eight XCDs interleave: linear = ((k / R) * 8 + xcd) * R +
// k % R.  A permutation of [0, full) where full = the workgroups in whole groups of 8 R (the rest stay as dispatched), so
// every (frame, tile) still has exactly one workgroup.  frame = linear / tiles by a host-made reciprocal (magic = ceil(2^32 /
// tiles), exact while linear * tiles < 2^32: the host leaves the mapping off otherwise).  Scalar arithmetic only.
// Measured on K1's traffic shape (tools/ubench/k1_dma_ubench.hip, profiles/r04h - r04k_*): runs of 16 - 64 tiles +2 ... +3.4
// points of the HBM peak over the plain order on two boxes, whole frames per XCD +1.3 ... +2.6, runs of 4 or of 12 / 24 / 48
// tiles little or nothing; in the kernels (same box, alternating): DESIGN.md section 5.
#ifdef __HIPCC__
__device__ __forceinline__ void xcd_work(int map, unsigned magic, unsigned &frame, unsigned &tile) {
    frame = blockIdx.y;
    tile = blockIdx.x;
    if (map > 0) {
        const unsigned per = gridDim.x, id = frame * per + tile, sh = (unsigned)map - 1u;
        const unsigned group = 8u << sh, total = per * gridDim.y;
        if (id < total - total % group) {
            const unsigned k = id >> 3;
            const unsigned lin = ((((k >> sh) << 3) + (id & 7u)) << sh) + (k & ((1u << sh) - 1u));
            frame = __umulhi(lin, magic);
            tile = lin - frame * per;
        }
    }
}
#endif
// host side: the reciprocal, and whether the mapping may be used for a grid of `per` x `n` workgroups (0 = off)
inline int xcd_map_for(unsigned per, unsigned n, unsigned &magic, bool fused = false) {
    magic = per > 1 ? (unsigned)(((1ull << 32) + per - 1) / per) : 0u;
    const int run = xcd_map_run(fused);
    if (run <= 0 || per <= 1 || (unsigned long long)per * n * per >= (1ull << 32)) return 0;
    int sh = 0;
    while ((1 << (sh + 1)) <= run) sh++;
    return sh + 1;
}

// k0/k1 (optional): events recorded right before / after the dominant kernel.
// wide_only: every block through the int64 kernel (tables with entries > 255).
hipError_t launch_decode_444(const Decode444Params &P, bool wide_only, hipStream_t s, hipEvent_t k0 = nullptr,
                             hipEvent_t k1 = nullptr);
// nw, y_tiles, c_tiles_*, magics, tiles_per_frame from the planes' crop geometry (P.pl[], P.width set); aligned = the
// 16-byte store form can be used (width % 16 == 0, frame stride % 16 == 0, 16-byte aligned output)
void plan_decode_444(Decode444Params &P, bool aligned);
hipError_t launch_decode(const DecodeParams &P, hipStream_t s, hipEvent_t k0 = nullptr, hipEvent_t k1 = nullptr);
hipError_t launch_decode_wide_only(const DecodeParams &P, hipStream_t s, hipEvent_t k0 = nullptr, hipEvent_t k1 = nullptr);
// after a batch's launches: the listed blocks (fix-list ids of P's geometry) recomputed in int64 with the DC of dcs[]
hipError_t launch_decode_dcfix(const DecodeParams &P, const unsigned *count, const unsigned *ids, const long long *dcs,
                               hipStream_t s);
hipError_t launch_decode_444_dcfix(const Decode444Params &P, const unsigned *count, const unsigned *ids, const long long *dcs,
                                   unsigned n_host, hipStream_t s);
hipError_t launch_encode(const EncodeParams &P, hipStream_t s, hipEvent_t k0 = nullptr, hipEvent_t k1 = nullptr);
hipError_t launch_upsample420(const UpsampleParams &P, hipStream_t s);
// error = |recon - P.pixels| per sample of the component planes (Encoder.recon, encoder.ml:119-125); P.coefs unused
hipError_t launch_abs_error(const EncodeParams &P, const uint8_t *recon, uint8_t *error, hipStream_t s);
// K5: sums[r] (device, n_records entries, cleared here) = position-weighted 64-bit checksum of record r
#define HVC_CHECKSUM_MUL 0x9E3779B97F4A7C15ull
hipError_t launch_checksum(const uint8_t *data, size_t record_bytes, size_t record_stride, int n_records,
                           unsigned long long *sums, hipStream_t s);

} // namespace hvc
#endif
