// hvc_huff.hip -- baseline Huffman coding of quantised coefficient records ON THE GPU (gfx950).
//
// The encoder's back end (Encoder.rle + write_bits + Bitstream_writer with byte stuffing,
// jpeg/model/src/encoder.ml:127-193, common/src/bitstream_writer.ml) is sequential in the model, but
// nothing in it depends on earlier OUTPUT: every block's bit string is a function of its own 64
// coefficients and of one other DC value (the predictor = the previous block of the same component
// in scan order).  So the scan is rebuilt as data-parallel passes over device-resident records:
//   1  k_huff_len     one block per lane: bit length of the block's code -> lens[scan index]
//   2  k_scan_u32     per frame exclusive prefix sum (one workgroup per frame)   -> bit offsets
//   3  k_huff_emit    one block per lane: the same walk, now writing the bits at the block's offset
//                     (big-endian bit order; words shared with a neighbour by atomicOr)
//   4  k_ff_count / k_scan_u32 / k_frame_offsets / k_stuff_write: 0xFF -> 0xFF 0x00 and packing of the
//                     frames' segments back to back
// The bytes equal hvc_jpeg_entropy_encode's scan data (tests/test_gpu_huffman.py), hence the model's.
// Default (Annex K) tables only, as Encoder.Parameters.c420/c422/c444 use (encoder.ml:306-349).
#include <vector>

#include "hvc_huff.h"

namespace hvc {

namespace {

constexpr int HT = 256; // blocks (lanes) per workgroup, one component plane tile like K1 / K3

struct BlockPos {
    int comp, bx, by;
    bool active;
    unsigned scan;     // index of the block in scan order inside its frame
    size_t coef_idx;   // int16 element index of the block's coefficients
    size_t pred_idx;   // ... of the block whose DC is the predictor (valid when has_pred)
    bool has_pred;
};

__device__ __forceinline__ BlockPos locate_block(const HuffParams &P, int frame, int tile, int lane) {
    BlockPos r;
    int c = 0;
#pragma unroll
    for (int i = 1; i < 3; i++)
        if (tile >= P.comp[i].tile0) c = i;
    const HuffComp &K = P.comp[c];
    int b = (tile - K.tile0) * HT + lane;
    r.active = b < K.nblk;
    b = r.active ? b : K.nblk - 1;
    const int by = b / K.bw, bx = b - by * K.bw;
    r.comp = c;
    r.bx = bx;
    r.by = by;
    const size_t base = (size_t)frame * P.coef_fs + K.coef_off;
    r.coef_idx = base + (size_t)b * 64;
    // scan order (encoder.ml:476-505): MCU rows, MCUs, components, v x h blocks inside the MCU
    const int mx = bx / K.h, sx = bx - mx * K.h, my = by / K.v, sy = by - my * K.v;
    // blocks outside the MCU grid (planes larger than the grid) are never coded
    if (mx >= P.mbs_wide || my >= P.mbs_high) r.active = false;
    r.scan = (unsigned)(my * P.mbs_wide + mx) * (unsigned)P.blocks_per_mcu + (unsigned)(K.mcu_base + sy * K.h + sx);
    // predictor: the block coded just before this one in the same component
    const int hv = K.h * K.v;
    const int ord = (my * P.mbs_wide + mx) * hv + sy * K.h + sx;
    r.has_pred = ord > 0;
    const int po = r.has_pred ? ord - 1 : 0;
    const int pm = po / hv, pr = po - pm * hv;
    const int psy = pr / K.h, psx = pr - psy * K.h;
    const int pmy = pm / P.mbs_wide, pmx = pm - pmy * P.mbs_wide;
    r.pred_idx = base + ((size_t)(pmy * K.v + psy) * K.bw + (size_t)(pmx * K.h + psx)) * 64;
    return r;
}

// The walk over one block, shared by the length and the emit pass.  SINK::put(code, len).
template <class SINK>
__device__ __forceinline__ void walk_block(const unsigned (&w)[32], int pred, const unsigned *tab /* LDS: 16 dc + 256 ac */,
                                           SINK &sink, unsigned &err) {
    const int dc = (int)(short)(w[0] & 0xffffu);
    const int diff = dc - pred;
    {
        const unsigned a = (unsigned)(diff < 0 ? -diff : diff);
        const int size = a ? 32 - __clz((int)a) : 0;
        if (size > 11) err = 1; // no code in the default DC tables (the host coder returns HVC_E_RANGE)
        const unsigned e = tab[size & 15];
        const unsigned mag = (unsigned)(diff >= 0 ? diff : diff - 1) & ((1u << size) - 1u);
        sink.put(((e >> 5) << size) | mag, (int)(e & 31u) + size);
    }
    int run = 0;
#pragma unroll
    for (int k = 1; k < 64; k++) {
        const int v = (k & 1) ? (int)w[k >> 1] >> 16 : (int)(short)(w[k >> 1] & 0xffffu);
        if (v == 0) {
            run++;
        } else {
            while (run >= 16) { // ZRL (encoder.ml:162-187)
                const unsigned z = tab[16 + 0xf0];
                sink.put(z >> 5, (int)(z & 31u));
                run -= 16;
            }
            const unsigned a = (unsigned)(v < 0 ? -v : v);
            const int size = 32 - __clz((int)a);
            if (size > 10) err = 1; // no code in the default AC tables
            const unsigned e = tab[16 + ((run << 4) | (size & 15))];
            const unsigned mag = (unsigned)(v >= 0 ? v : v - 1) & ((1u << size) - 1u);
            sink.put(((e >> 5) << size) | mag, (int)(e & 31u) + size);
            run = 0;
        }
    }
    if (run) { // EOB
        const unsigned z = tab[16];
        sink.put(z >> 5, (int)(z & 31u));
    }
}

struct LenSink {
    unsigned bits = 0;
    __device__ __forceinline__ void put(unsigned, int len) { bits += (unsigned)len; }
};

// Writes a bit string at an arbitrary bit offset of a zero-initialised big-endian bit buffer.  The
// first and the last word of the string may be shared with the neighbouring blocks: atomicOr; the
// words in between belong to this block alone: plain stores.
struct EmitSink {
    unsigned *wp;            // next 32-bit word of the frame's buffer
    unsigned long long acc;  // pending bits, right-aligned
    int n;                   // number of pending bits (including the `lead` bits of the first word)
    bool first;
    __device__ __forceinline__ void init(unsigned *buf, unsigned long long bitpos) {
        wp = buf + (bitpos >> 5);
        acc = 0;
        n = (int)(bitpos & 31u); // the leading bits of the first word are somebody else's: zeros here
        first = n != 0;
    }
    __device__ __forceinline__ void put(unsigned code, int len) { // len <= 27
        acc = (acc << len) | code;
        n += len;
        if (n >= 32) {
            const unsigned word = (unsigned)(acc >> (n - 32));
            n -= 32;
            const unsigned be = __builtin_bswap32(word);
            if (first)
                atomicOr(wp, be);
            else
                *wp = be;
            first = false;
            wp++;
        }
    }
    __device__ __forceinline__ void finish() {
        if (n > 0) atomicOr(wp, __builtin_bswap32((unsigned)(acc << (32 - n))));
    }
};

__device__ __forceinline__ void load_tables(const HuffParams &P, unsigned *lds) {
    for (int i = threadIdx.x; i < 2 * 272; i += HT) lds[i] = P.tables[i]; // both table sets
}

} // namespace

// pass 1 ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(HT) void k_huff_len(HuffParams P) {
    __shared__ unsigned tabs[2 * 272];
    load_tables(P, tabs);
    __syncthreads();
    const int lane = threadIdx.x, frame = blockIdx.y;
    const BlockPos b = locate_block(P, frame, blockIdx.x, lane);
    const uint4 *src = reinterpret_cast<const uint4 *>(P.coefs + b.coef_idx);
    unsigned w[32];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const uint4 t = src[j];
        w[4 * j + 0] = t.x;
        w[4 * j + 1] = t.y;
        w[4 * j + 2] = t.z;
        w[4 * j + 3] = t.w;
    }
    const int pred = b.has_pred ? (int)P.coefs[b.pred_idx] : 0;
    LenSink s;
    unsigned err = 0;
    walk_block(w, pred, tabs + 272 * P.comp[b.comp].table, s, err);
    if (b.active) {
        P.lens[(size_t)frame * P.blocks_per_frame + b.scan] = s.bits;
        if (err) atomicOr(P.status, 1u);
    }
}

// pass 3 ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(HT) void k_huff_emit(HuffParams P) {
    __shared__ unsigned tabs[2 * 272];
    load_tables(P, tabs);
    __syncthreads();
    const int lane = threadIdx.x, frame = blockIdx.y;
    const BlockPos b = locate_block(P, frame, blockIdx.x, lane);
    if (!b.active) return;
    if ((size_t)((P.frame_bits[frame] + 31u) >> 5) + 1 > P.bitbuf_words) return; // flagged by k_frame_sizes
    const uint4 *src = reinterpret_cast<const uint4 *>(P.coefs + b.coef_idx);
    unsigned w[32];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const uint4 t = src[j];
        w[4 * j + 0] = t.x;
        w[4 * j + 1] = t.y;
        w[4 * j + 2] = t.z;
        w[4 * j + 3] = t.w;
    }
    const int pred = b.has_pred ? (int)P.coefs[b.pred_idx] : 0;
    const size_t li = (size_t)frame * P.blocks_per_frame + b.scan;
    const unsigned bitpos = P.lens[li]; // exclusive offset after pass 2
    EmitSink s;
    s.init(P.bitbuf + (size_t)frame * P.bitbuf_words, bitpos);
    unsigned err = 0;
    walk_block(w, pred, tabs + 272 * P.comp[b.comp].table, s, err);
    if (b.scan == P.blocks_per_frame - 1) {
        // Bitstream_writer.flush_with_1s (bitstream_writer.ml:45-49): pad the last byte with ones
        const unsigned total = P.frame_bits[frame];
        const int pad = (int)((8u - (total & 7u)) & 7u);
        if (pad) s.put((1u << pad) - 1u, pad);
    }
    s.finish();
}

// pass 2 / 4b: per-frame exclusive scan of n[frame] 32-bit values (in place), total -> totals[frame].
// One workgroup of 1024 lanes per frame; 4 values per lane per round.
__global__ __launch_bounds__(1024) void k_scan_u32(unsigned *data, size_t stride, const unsigned *counts, unsigned fixed_count,
                                                   unsigned *totals) {
    __shared__ unsigned wsum[16];
    __shared__ unsigned carry_s;
    const int frame = blockIdx.x, lane = threadIdx.x, wave = lane >> 6, wl = lane & 63;
    unsigned *d = data + (size_t)frame * stride;
    const unsigned n = counts ? counts[frame] : fixed_count;
    if (lane == 0) carry_s = 0;
    __syncthreads();
    for (unsigned base = 0; base < n; base += 4096) {
        const unsigned i0 = base + 4u * (unsigned)lane;
        unsigned v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = (i0 + k < n) ? d[i0 + k] : 0u;
        const unsigned mine = v[0] + v[1] + v[2] + v[3];
        // inclusive scan of `mine` inside the wave
        unsigned incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned t = __shfl_up(incl, o);
            if (wl >= o) incl += t;
        }
        if (wl == 63) wsum[wave] = incl;
        __syncthreads();
        unsigned wbase = 0;
        for (int k = 0; k < wave; k++) wbase += wsum[k];
        unsigned run = carry_s + wbase + incl - mine;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (i0 + k < n) d[i0 + k] = run;
            run += v[k];
        }
        __syncthreads();
        if (lane == 1023) carry_s = run; // run of the last lane = carry + everything of this round
        __syncthreads();
    }
    if (lane == 0) totals[frame] = carry_s;
}

// After pass 2: bytes of every frame's unstuffed segment and the number of 64-byte pieces.
__global__ void k_frame_sizes(HuffParams P) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= P.n_frames) return;
    const unsigned bits = P.frame_bits[f];
    const unsigned bytes = (bits + 7u) >> 3;
    P.frame_bytes[f] = bytes;
    P.frame_pieces[f] = (bytes + 63u) >> 6;
    if ((size_t)((bits + 31u) >> 5) + 1 > P.bitbuf_words) atomicOr(P.status, 2u); // cannot happen: worst case sized
}

// pass 4a: number of 0xFF bytes in every 64-byte piece
__global__ __launch_bounds__(256) void k_ff_count(HuffParams P) {
    const int frame = blockIdx.y;
    const unsigned piece = blockIdx.x * 256u + threadIdx.x;
    if (piece >= P.frame_pieces[frame]) return;
    const unsigned bytes = P.frame_bytes[frame];
    const uint4 *src = reinterpret_cast<const uint4 *>(P.bitbuf + (size_t)frame * P.bitbuf_words) + (size_t)piece * 4;
    unsigned cnt = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint4 t = src[j];
        const unsigned ws[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const unsigned pos = piece * 64u + (unsigned)(j * 16 + k * 4);
#pragma unroll
            for (int bb = 0; bb < 4; bb++)
                cnt += (pos + bb < bytes && ((ws[k] >> (8 * bb)) & 0xffu) == 0xffu) ? 1u : 0u;
        }
    }
    P.ff[(size_t)frame * P.ff_stride + piece] = cnt;
}

// pass 4c: offsets of the frames' stuffed segments inside the packed output (n_frames is small)
__global__ void k_frame_offsets(HuffParams P) {
    if (blockIdx.x || threadIdx.x) return;
    unsigned long long off = 0;
    for (int f = 0; f < P.n_frames; f++) {
        P.out_offsets[f] = off;
        off += (unsigned long long)P.frame_bytes[f] + P.frame_ff[f];
    }
    P.out_offsets[P.n_frames] = off;
    if (off > P.out_cap) atomicOr(P.status, 4u);
}

// pass 4d: copy with stuffing; one 64-byte piece per lane
__global__ __launch_bounds__(256) void k_stuff_write(HuffParams P) {
    const int frame = blockIdx.y;
    const unsigned piece = blockIdx.x * 256u + threadIdx.x;
    if (piece >= P.frame_pieces[frame]) return;
    if (P.out_offsets[P.n_frames] > P.out_cap) return;
    const unsigned bytes = P.frame_bytes[frame];
    const uint8_t *src = reinterpret_cast<const uint8_t *>(P.bitbuf + (size_t)frame * P.bitbuf_words) + (size_t)piece * 64;
    uint8_t *dst = P.out + P.out_offsets[frame] + (size_t)piece * 64 + P.ff[(size_t)frame * P.ff_stride + piece];
    const unsigned n = min(64u, bytes - piece * 64u);
    const unsigned nff = (piece + 1 < P.frame_pieces[frame] ? P.ff[(size_t)frame * P.ff_stride + piece + 1]
                                                            : P.frame_ff[frame]) -
                         P.ff[(size_t)frame * P.ff_stride + piece];
    if (nff == 0 && n == 64) {
        // four out of five pieces hold no 0xFF: a straight copy, dwords at whatever alignment the
        // earlier stuffing bytes left (global memory takes unaligned dwords)
        typedef unsigned unaligned_u32 __attribute__((aligned(1)));
        const uint4 *s4 = reinterpret_cast<const uint4 *>(src);
        unaligned_u32 *d = reinterpret_cast<unaligned_u32 *>(dst);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint4 t = s4[j];
            d[4 * j + 0] = t.x;
            d[4 * j + 1] = t.y;
            d[4 * j + 2] = t.z;
            d[4 * j + 3] = t.w;
        }
        return;
    }
    for (unsigned i = 0; i < n; i++) {
        const uint8_t v = src[i];
        *dst++ = v;
        if (v == 0xff) *dst++ = 0;
    }
}

hipError_t launch_huffman_encode(const HuffParams &P, hipStream_t s) {
    if (P.n_frames <= 0) return hipSuccess;
    hipError_t e = hipMemsetAsync(P.status, 0, sizeof(unsigned), s);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(P.bitbuf, 0, (size_t)P.n_frames * P.bitbuf_words * sizeof(unsigned), s);
    if (e != hipSuccess) return e;
    const dim3 grid((unsigned)P.tiles_per_frame, (unsigned)P.n_frames, 1);
    hipLaunchKernelGGL(k_huff_len, grid, dim3(HT), 0, s, P);
    hipLaunchKernelGGL(k_scan_u32, dim3((unsigned)P.n_frames), dim3(1024), 0, s, P.lens, (size_t)P.blocks_per_frame,
                       (const unsigned *)nullptr, P.blocks_per_frame, P.frame_bits);
    hipLaunchKernelGGL(k_frame_sizes, dim3((unsigned)((P.n_frames + 63) / 64)), dim3(64), 0, s, P);
    hipLaunchKernelGGL(k_huff_emit, grid, dim3(HT), 0, s, P);
    const unsigned max_pieces = (unsigned)((P.bitbuf_words * 4 + 63) / 64);
    const dim3 pgrid((max_pieces + 255u) / 256u, (unsigned)P.n_frames, 1);
    hipLaunchKernelGGL(k_ff_count, pgrid, dim3(256), 0, s, P);
    hipLaunchKernelGGL(k_scan_u32, dim3((unsigned)P.n_frames), dim3(1024), 0, s, P.ff, P.ff_stride,
                       (const unsigned *)P.frame_pieces, 0u, P.frame_ff);
    hipLaunchKernelGGL(k_frame_offsets, dim3(1), dim3(1), 0, s, P);
    hipLaunchKernelGGL(k_stuff_write, pgrid, dim3(256), 0, s, P);
    return hipGetLastError();
}

} // namespace hvc
