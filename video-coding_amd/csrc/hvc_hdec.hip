// hvc_hdec.hip -- baseline Huffman DEcoding of entropy-coded segments on the GPU (gfx950).
//
// The model's reader (Decoder.huffman_decode, jpeg/model/src/decoder.ml:118-140, over Bits) is a
// sequential walk: where a symbol starts is known only after the previous one has been decoded.
// Huffman streams re-synchronise, though: a decoder started at a wrong position (and in a wrong
// state) falls into step with the true parse after a while.  The segment is therefore cut into
// subsequences of HVC_HD_SUBSEQ_BITS bits, one lane each:
//   round 0      every lane decodes its subsequence from its first bit with a guessed state
//                (expecting a DC symbol of the first block of an MCU) and records where and in which
//                state (bit position, zig-zag index, block inside the MCU) it leaves it
//   round r > 0  lane i restarts from the exit of lane i - 1 of the previous round -- if that differs from
//                what it started from last time; lane 0 always had the true start, so after round r the
//                first r + 1 exits are true, and in practice all of them are after a handful of rounds.
//                The host stops when a round changed nothing.
//   finish       blocks completed per subsequence -> exclusive scan = index of the block each lane starts
//                in; every lane decodes once more, now writing coefficients (zig-zag order, DC as the
//                DIFFERENCE); one workgroup per component turns the DC differences into values by a
//                prefix sum in scan order (decoder.ml:143), checking the int16 range.
// Anything the model would raise on (invalid code, index past 63, DC category > 16 inside the coded
// blocks), a DC outside int16 or a stream that ends before the frame is complete only raises a status
// bit here: the caller then runs the host decoder, which reproduces the model's behaviour exactly.
#include "hvc_hdec.h"

namespace hvc {

namespace {

constexpr int S = HVC_HD_SUBSEQ_BITS;

__device__ __forceinline__ unsigned long long pack_state(unsigned p, int k, int b) {
    return (unsigned long long)p | ((unsigned long long)(unsigned)k << 32) | ((unsigned long long)(unsigned)b << 40);
}

// 64 bits of the stream starting at bit position p (MSB first), from a zero-padded byte buffer
__device__ __forceinline__ unsigned long long window(const uint8_t *ecs, unsigned p) {
    typedef unsigned unaligned_u32 __attribute__((aligned(1)));
    const uint8_t *q = ecs + (p >> 3);
    const unsigned a = __builtin_bswap32(*reinterpret_cast<const unaligned_u32 *>(q));
    const unsigned b = __builtin_bswap32(*reinterpret_cast<const unaligned_u32 *>(q + 4));
    const unsigned c = __builtin_bswap32(*reinterpret_cast<const unaligned_u32 *>(q + 8));
    const unsigned long long hi = ((unsigned long long)a << 32) | b;
    const int sh = (int)(p & 7u);
    return sh ? (hi << sh) | ((unsigned long long)c >> (32 - sh)) : hi;
}

// (length << 8) | value, 0 = no code.  w = the next 64 bits.
__device__ __forceinline__ unsigned lookup(const HdTable &t, unsigned long long w) {
    unsigned e = t.fast[(unsigned)(w >> 54)];
    if (e) return e;
    for (int len = 11; len <= t.max_bits; len++) {
        const unsigned code = (unsigned)(w >> (64 - len));
        const unsigned d = code - t.first[len];
        if (code >= t.first[len] && d < t.count[len]) return ((unsigned)len << 8) | t.vals[t.voff[len] + d];
    }
    return 0;
}

__device__ __forceinline__ int extend(int cat, unsigned code) { // decoder.ml:73-79 mag'
    return (code & (1u << (cat - 1))) ? (int)code : (int)code - (int)((1u << cat) - 1);
}

// WRITE = false: walk only.  WRITE = true: store coefficients of blocks [0, blocks_per_frame).
// Decodes symbols while p < limit.  Returns the exit state through p, k, b and the number of blocks
// completed in nb.  err: bit 0 set when the walk hits something the model raises on.
template <bool WRITE>
__device__ __forceinline__ void walk(const HdParams &P, const HdTables &T, const uint8_t *ecs, unsigned limit, unsigned &p,
                                     int &k, int &b, unsigned &nb, unsigned first_block, int16_t *rec, unsigned &err) {
    const int B = P.blocks_per_mcu;
    unsigned bi = first_block;
    int16_t *blk = nullptr;
    auto block_ptr = [&](unsigned index, int bb) -> int16_t * {
        const unsigned mcu = index / (unsigned)B;
        const int comp = P.b2comp[bb];
        const HdComp &C = P.comp[comp];
        const int r = bb - C.mcu_base, sy = r / C.h, sx = r - sy * C.h;
        const unsigned my = mcu / (unsigned)P.mbs_wide, mx = mcu - my * (unsigned)P.mbs_wide;
        return rec + C.coef_off + ((size_t)(my * C.v + sy) * C.bw + (size_t)(mx * C.h + sx)) * 64;
    };
    if (WRITE && bi < P.blocks_per_frame) blk = block_ptr(bi, b);
    while (p < limit) {
        const unsigned long long w = window(ecs, p);
        const int comp = P.b2comp[b];
        bool end_block = false;
        if (k == 0) {
            const unsigned e = lookup(T.dc[comp], w);
            if (!e) { // "Can't find dc code": a real error in the true parse, noise in a speculative one
                if (WRITE && bi < P.blocks_per_frame) err |= 1u;
                p += 1;
                continue;
            }
            const int len = (int)(e >> 8), cat = (int)(e & 0xffu);
            if (cat > 16) {
                if (WRITE && bi < P.blocks_per_frame) err |= 1u;
                p += (unsigned)len;
                continue;
            }
            int diff = 0;
            if (cat) diff = extend(cat, (unsigned)((w << len) >> (64 - cat)));
            p += (unsigned)(len + cat);
            if (WRITE && bi < P.blocks_per_frame) {
                // the difference; k_hd_dc turns it into the value.  A category above 15 cannot be an int16: flagged there
                blk[0] = (int16_t)diff;
                if (diff < -32768 || diff > 32767) err |= 2u;
            }
            k = 1;
        } else {
            const unsigned e = lookup(T.ac[comp], w);
            if (!e) { // "Can't find ac code"
                if (WRITE && bi < P.blocks_per_frame) err |= 1u;
                p += 1;
                continue;
            }
            const int len = (int)(e >> 8), run = (int)((e >> 4) & 15u), size = (int)(e & 15u);
            int mag = 0;
            if (size) mag = extend(size, (unsigned)((w << len) >> (64 - size)));
            p += (unsigned)(len + size);
            if (mag == 0 && run == 0) { // EOB (or any zero-size code with run 0), decoder.ml:131-132
                end_block = true;
            } else {
                k += run;
                if (k >= 64) { // "coefficient index out of range"
                    if (WRITE && bi < P.blocks_per_frame) err |= 1u;
                    end_block = true;
                } else {
                    if (WRITE && bi < P.blocks_per_frame && mag) blk[k] = (int16_t)mag;
                    k++;
                    if (k == 64) end_block = true;
                }
            }
        }
        if (end_block) {
            k = 0;
            b = b + 1 == B ? 0 : b + 1;
            nb++;
            bi++;
            if (WRITE && bi < P.blocks_per_frame) blk = block_ptr(bi, b);
        }
    }
}

} // namespace

// One synchronisation round (see the header comment).  Even rounds write exit_a, odd rounds exit_b.
__global__ __launch_bounds__(256) void k_hd_round(HdParams P, int round) {
    __shared__ HdTables T;
    {
        const unsigned *src = reinterpret_cast<const unsigned *>(P.tables);
        unsigned *dst = reinterpret_cast<unsigned *>(&T);
        for (unsigned i = threadIdx.x; i < sizeof(HdTables) / 4; i += 256) dst[i] = src[i];
    }
    __syncthreads();
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= P.total_sub) return;
    const unsigned f = P.frame_of[i], j = i - P.sub_off[f];
    const unsigned long long *prev = (round & 1) ? P.exit_a : P.exit_b;
    unsigned long long *cur = (round & 1) ? P.exit_b : P.exit_a;
    unsigned long long st;
    if (round == 0 || j == 0)
        st = pack_state(j * (unsigned)S, 0, 0);
    else
        st = prev[i - 1];
    if (round > 0 && st == P.start_used[i]) {
        cur[i] = prev[i];
        return;
    }
    unsigned p = (unsigned)st;
    int k = (int)((st >> 32) & 0xffu), b = (int)((st >> 40) & 0xffu);
    unsigned nb = 0, err = 0;
    walk<false>(P, T, P.ecs + P.ecs_off[f], (j + 1) * (unsigned)S, p, k, b, nb, 0, nullptr, err);
    cur[i] = pack_state(p, k, b);
    P.start_used[i] = st;
    P.nblk[i] = nb;
    if (round > 0) *P.changed = 1u; // benign race: every writer stores the same value
}

// Exclusive scan of nblk inside every frame (one workgroup per frame); total -> frame_blocks.
__global__ __launch_bounds__(1024) void k_hd_scan(HdParams P) {
    __shared__ unsigned wsum[16];
    __shared__ unsigned carry_s;
    const int frame = blockIdx.x, lane = threadIdx.x, wave = lane >> 6, wl = lane & 63;
    unsigned *d = P.nblk + P.sub_off[frame];
    const unsigned n = P.sub_off[frame + 1] - P.sub_off[frame];
    if (lane == 0) carry_s = 0;
    __syncthreads();
    for (unsigned base = 0; base < n; base += 1024) {
        const unsigned idx = base + (unsigned)lane;
        const unsigned v = idx < n ? d[idx] : 0u;
        unsigned incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned t = __shfl_up(incl, o);
            if (wl >= o) incl += t;
        }
        if (wl == 63) wsum[wave] = incl;
        __syncthreads();
        unsigned wbase = 0;
        for (int q = 0; q < wave; q++) wbase += wsum[q];
        const unsigned excl = carry_s + wbase + incl - v;
        if (idx < n) d[idx] = excl;
        __syncthreads();
        if (lane == 1023) carry_s = excl + v;
        __syncthreads();
    }
    if (lane == 0) {
        P.frame_blocks[frame] = carry_s;
        if (carry_s < P.blocks_per_frame) atomicOr(P.status, 4u); // the stream ends before the frame does
    }
}

// The write pass: every lane decodes its subsequence from its (now true) start and stores coefficients.
__global__ __launch_bounds__(256) void k_hd_write(HdParams P, int final_round) {
    __shared__ HdTables T;
    {
        const unsigned *src = reinterpret_cast<const unsigned *>(P.tables);
        unsigned *dst = reinterpret_cast<unsigned *>(&T);
        for (unsigned i = threadIdx.x; i < sizeof(HdTables) / 4; i += 256) dst[i] = src[i];
    }
    __syncthreads();
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= P.total_sub) return;
    const unsigned f = P.frame_of[i], j = i - P.sub_off[f];
    const unsigned first_block = P.nblk[i];
    if (first_block >= P.blocks_per_frame) return; // past the last coded block: the model never reads this far
    const unsigned long long st = P.start_used[i];
    unsigned p = (unsigned)st;
    int k = (int)((st >> 32) & 0xffu), b = (int)((st >> 40) & 0xffu);
    unsigned nb = 0, err = 0;
    walk<true>(P, T, P.ecs + P.ecs_off[f], (j + 1) * (unsigned)S, p, k, b, nb, first_block,
               P.coefs + (size_t)f * P.coef_fs, err);
    if (err) atomicOr(P.status, err);
    (void)final_round;
}

// DC differences -> DC values (decoder.ml:143): inclusive prefix sum over the component's blocks in scan
// order, one workgroup per (component, frame).
__global__ __launch_bounds__(1024) void k_hd_dc(HdParams P) {
    __shared__ int wsum[16];
    __shared__ int carry_s;
    const int comp = blockIdx.x, frame = blockIdx.y, lane = threadIdx.x, wave = lane >> 6, wl = lane & 63;
    if (comp >= P.n_comp) return;
    const HdComp &C = P.comp[comp];
    const int hv = C.h * C.v;
    const unsigned n = (unsigned)P.mbs_wide * (unsigned)P.mbs_high * (unsigned)hv;
    int16_t *rec = P.coefs + (size_t)frame * P.coef_fs + C.coef_off;
    if (lane == 0) carry_s = 0;
    __syncthreads();
    bool bad = false;
    for (unsigned base = 0; base < n; base += 1024) {
        const unsigned o = base + (unsigned)lane;
        int16_t *dcp = nullptr;
        int v = 0;
        if (o < n) {
            const unsigned m = o / (unsigned)hv, r = o - m * (unsigned)hv;
            const unsigned sy = r / (unsigned)C.h, sx = r - sy * (unsigned)C.h;
            const unsigned my = m / (unsigned)P.mbs_wide, mx = m - my * (unsigned)P.mbs_wide;
            dcp = rec + ((size_t)(my * C.v + sy) * C.bw + (size_t)(mx * C.h + sx)) * 64;
            v = *dcp;
        }
        int incl = v;
#pragma unroll
        for (int s = 1; s < 64; s <<= 1) {
            const int t = __shfl_up(incl, s);
            if (wl >= s) incl += t;
        }
        if (wl == 63) wsum[wave] = incl;
        __syncthreads();
        int wbase = 0;
        for (int q = 0; q < wave; q++) wbase += wsum[q];
        const int dc = carry_s + wbase + incl;
        if (o < n) {
            if (dc < -32768 || dc > 32767) bad = true;
            *dcp = (int16_t)dc;
        }
        __syncthreads();
        if (lane == 1023) carry_s = dc;
        __syncthreads();
    }
    if (bad) atomicOr(P.status, 2u);
}

hipError_t launch_hd_round(const HdParams &P, int round, hipStream_t s) {
    if (P.total_sub == 0) return hipSuccess;
    hipLaunchKernelGGL(k_hd_round, dim3((P.total_sub + 255u) / 256u), dim3(256), 0, s, P, round);
    return hipGetLastError();
}

hipError_t launch_hd_finish(const HdParams &P, int rounds_done, hipStream_t s) {
    if (P.total_sub == 0) return hipSuccess;
    hipLaunchKernelGGL(k_hd_scan, dim3((unsigned)P.n_frames), dim3(1024), 0, s, P);
    hipLaunchKernelGGL(k_hd_write, dim3((P.total_sub + 255u) / 256u), dim3(256), 0, s, P, rounds_done);
    hipLaunchKernelGGL(k_hd_dc, dim3((unsigned)P.n_comp, (unsigned)P.n_frames), dim3(1024), 0, s, P);
    return hipGetLastError();
}

} // namespace hvc
