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
//
// Two sets of kernels do this.  The general one -- k_hd_round (rounds inside a workgroup through LDS,
// across workgroups through launches) and k_hd_write -- takes any frame the host side lets through.
// The fast one -- k_hd_sync (level-synchronous rounds over work lists, skip-only tables, subsequences
// staged in LDS; k_hd_sync_tail: the late rounds of a few files inside one workgroup) and k_hd_write2
// (blocks owned by the lane they start in, wavefront-wide batched stores) -- needs the components to share at most
// two (DC, AC) table pairs, as every baseline file's do, or runs with per-frame tables from device memory (PF);
// launch_hd_round / launch_hd_finish pick it whenever HdParams::spec or ::ftabs is set, and k_hd_round then only
// verifies the hand-overs and finishes what takes more rounds than k_hd_sync is given.
#include "hvc_hdec.h"

#include <cstdlib>
#include <cstring>
#include <type_traits>

namespace hvc {

namespace {

constexpr int S = HVC_HD_SUBSEQ_BITS;

// After `rounds_done` synchronisation launches (0 .. rounds_done - 1): did the last one still change something?
// k_hd_round stamps *P.changed with the number of the launch that changed something (launch 0 never does).
__device__ __forceinline__ bool hd_unsettled(const HdParams &P, int rounds_done) {
    return rounds_done > 1 && *P.changed == (unsigned)(rounds_done - 1);
}

__device__ __forceinline__ unsigned long long pack_state(unsigned p, int k, int b) {
    return (unsigned long long)p | ((unsigned long long)(unsigned)k << 32) | ((unsigned long long)(unsigned)b << 40);
}

// Frame f of the reader as a place in the batch's records (HdParams::rst_*): the file whose record its blocks go to, the
// first of the file's MCUs it holds, how many blocks it has.  Without restart intervals: the frame itself, 0, all.
struct FrameRef {
    unsigned file, mcu0, need;
};
__device__ __forceinline__ FrameRef frame_ref(const HdParams &P, unsigned f) {
    FrameRef r{f, 0u, P.blocks_per_frame};
    if (P.rst_ipf > 1u) {
        r.file = f / P.rst_ipf;
        r.mcu0 = (f - r.file * P.rst_ipf) * P.rst_mcus;
        const unsigned left = (unsigned)P.mbs_wide * (unsigned)P.mbs_high - r.mcu0; // (>= 1: rst_ipf = ceil(MCUs / rst_mcus))
        r.need = (left < P.rst_mcus ? left : P.rst_mcus) * (unsigned)P.blocks_per_mcu;
    }
    return r;
}

// frames f0 <= f1: of one FILE -- one set of per-file tables (PF mode: a workgroup whose subsequences lie between them can
// keep that set in LDS)?  Frames are files, or with restart intervals a file's intervals, one after the other.
__device__ __forceinline__ bool one_file(const HdParams &P, unsigned f0, unsigned f1) {
    return f0 == f1 || (P.rst_ipf > 1u && f0 / P.rst_ipf == f1 / P.rst_ipf);
}

// A lane's bits come straight from global memory, one (byte-swapped) dword per 32 bits consumed, requested
// two refills ahead -- a dozen symbols -- so their latency is covered.  (A first version staged each lane's
// 144 bytes in LDS: 38 KB per workgroup, which held the kernel at 2 waves per SIMD; without it LDS holds only
// the tables and the CU runs four times as many waves of this latency-bound loop.)
__device__ __forceinline__ unsigned be32(const unsigned *g, unsigned i) { return __builtin_bswap32(g[i]); }

// (length << 8) | value, 0 = no code.  w = the next 64 bits.
__device__ __forceinline__ unsigned lookup(const HdTable &t, unsigned long long w) {
    unsigned e = t.fast[(unsigned)(w >> 54)];
    if (e & 0x8000u) e = t.sub[(e & 0x7fffu) * 64u + ((unsigned)(w >> 48) & 63u)];
    return e;
}

__device__ __forceinline__ int extend(int cat, unsigned code) { // decoder.ml:73-79 mag'
    return (code & (1u << (cat - 1))) ? (int)code : (int)code - (int)((1u << cat) - 1);
}

// The lane-varying part of the geometry, copied to LDS once per workgroup: indexing the kernel
// argument block with a per-lane index makes every access a load from the kernarg segment (host memory:
// ~1.5 us each -- it was 90 % of the first version's time).
struct HdGeo {
    int h[4], v[4], bw[4], mcu_base[4];
    unsigned coef_off[4];
    unsigned char b2comp[HVC_HD_MAX_MCU_BLOCKS], b2sx[HVC_HD_MAX_MCU_BLOCKS], b2sy[HVC_HD_MAX_MCU_BLOCKS];
};
__device__ __forceinline__ void load_geo(const HdParams &P, HdGeo &G) {
    if (threadIdx.x < 4) {
        const int i = threadIdx.x;
        G.h[i] = P.comp[i].h;
        G.v[i] = P.comp[i].v;
        G.bw[i] = P.comp[i].bw;
        G.mcu_base[i] = P.comp[i].mcu_base;
        G.coef_off[i] = (unsigned)P.comp[i].coef_off;
    }
    if (threadIdx.x < HVC_HD_MAX_MCU_BLOCKS) {
        const int bb = threadIdx.x, comp = P.b2comp[bb];
        const int r = bb - P.comp[comp].mcu_base, hh = P.comp[comp].h > 0 ? P.comp[comp].h : 1;
        G.b2comp[bb] = (unsigned char)comp;
        G.b2sy[bb] = (unsigned char)(r >= 0 ? r / hh : 0);
        G.b2sx[bb] = (unsigned char)(r >= 0 ? r % hh : 0);
    }
}

// WRITE = false: walk only.  WRITE = true: store coefficients of blocks [0, blocks_per_frame).
// Decodes symbols while p < limit.  Returns the exit state through p, k, b and the number of blocks
// completed in nb.  err: bit 0 set when the walk hits something the model raises on.
template <bool WRITE>
__device__ __forceinline__ void walk(const HdParams &P, const HdGeo &G, const HdTables &T, const unsigned *slot, unsigned base, unsigned limit,
                                     unsigned &p, int &k, int &b, unsigned &nb, unsigned first_block, int16_t *rec,
                                     unsigned &err, int16_t *lb = nullptr, unsigned mcu0 = 0u, unsigned need = 0u) {
    // WRITE: lb = this lane's 64-coefficient LDS buffer (zeroed).  Coefficients are assembled there and
    // leave as whole 128-byte blocks (eight 16-byte stores); storing them one by one -- 2 bytes at random
    // places of a record that is not in any cache -- made the write pass cost as much as all the
    // synchronisation rounds together.  A block that straddles two subsequences is written by both
    // lanes, each its own index range [lo, hi): the lane that decoded the DC owns [0, k at its exit), the
    // next lane the rest; nothing is cleared beforehand, every index of every coded block is written once.
    const int B = P.blocks_per_mcu;
    unsigned bi = first_block;
    int16_t *blk = nullptr;
    int lo = k; // first index of the current block this lane is responsible for
    // position of the current block: MCU coordinates advance by counting, no divisions inside the loop
    unsigned mx = 0, my = 0;
    if (WRITE) {
        const unsigned mcu = first_block / (unsigned)B + mcu0;
        my = mcu / (unsigned)P.mbs_wide;
        mx = mcu - my * (unsigned)P.mbs_wide;
    }
    auto block_ptr = [&](int bb) -> int16_t * {
        const int comp = G.b2comp[bb];
        return rec + G.coef_off[comp] +
               ((size_t)(my * (unsigned)G.v[comp] + G.b2sy[bb]) * (unsigned)G.bw[comp] + (size_t)(mx * (unsigned)G.h[comp] + G.b2sx[bb])) * 64;
    };
    bool live = WRITE && bi < need; // this block's coefficients are stored (and its errors count)
    if (live) blk = block_ptr(b);
    // The serial chain per symbol is what bounds the whole decoder, so it is kept short: a 64-bit
    // MSB-aligned window in registers (refilled a dword at a time from the lane's LDS slot, the next
    // dword already loaded), one table load, and no per-symbol geometry look-ups (component and
    // table pointers change only at block ends).
    unsigned off = p - base;               // bits consumed from the subsequence so far
    unsigned wi = (off >> 5) + 2;          // next dword to append (the subsequence's bytes + 16 of overshoot)
    unsigned long long buf = (((unsigned long long)be32(slot, off >> 5) << 32) | be32(slot, (off >> 5) + 1)) << (off & 31u);
    int avail = 64 - (int)(off & 31u);
    unsigned n0 = be32(slot, wi), n1 = be32(slot, min(wi + 1, (unsigned)(S / 32 + 3)));
    int comp = G.b2comp[b];
    const HdTable *dct = &T.dc[comp], *act = &T.ac[comp];
    while (p < limit) {
        if (avail <= 32) { // append a pre-loaded dword; request the one after next (indices stay inside the subsequence + overshoot)
            buf |= (unsigned long long)n0 << (32 - avail);
            avail += 32;
            wi++;
            n0 = n1;
            n1 = be32(slot, min(wi + 1, (unsigned)(S / 32 + 3)));
        }
        const HdTable &t = k ? *act : *dct;
        const unsigned e = lookup(t, buf);
        if (!e) { // "Can't find dc / ac code": a real error in the true parse, noise in a speculative one
            if (live) err |= 1u;
            buf <<= 1;
            avail -= 1;
            p += 1;
            continue;
        }
        const int len = (int)(e >> 8), val = (int)(e & 0xffu);
        const bool is_dc = k == 0;
        const int size = is_dc ? val : (val & 15), run = is_dc ? 0 : (val >> 4);
        if (is_dc && size > 16) { // DC category above 16
            if (live) err |= 1u;
            buf <<= len;
            avail -= len;
            p += (unsigned)len;
            continue;
        }
        // the value itself matters only when it is stored; the walk needs "is it zero", and a coded magnitude of
        // size > 0 never is (decoder.ml:73-79)
        int mag = size;
        if (WRITE && size) mag = extend(size, (unsigned)((buf << len) >> (64 - size)));
        const int used = len + size; // <= 32
        buf <<= used;
        avail -= used;
        p += (unsigned)used;
        bool end_block = false;
        if (is_dc) {
            if (live) { // the difference; k_hd_dc turns it into the value
                lb[0] = (int16_t)mag;
                if (mag < -32768 || mag > 32767) err |= 2u;
            }
            k = 1;
        } else if (mag == 0 && run == 0) { // EOB (or any zero-size code with run 0), decoder.ml:131-132
            end_block = true;
        } else {
            k += run;
            if (k >= 64) { // "coefficient index out of range"
                if (live) err |= 1u;
                end_block = true;
            } else {
                if (live && mag) lb[k] = (int16_t)mag;
                k++;
                end_block = k == 64;
            }
        }
        if (end_block) {
            if (live) { // the block is complete: flush [lo, 64) and clear the buffer
                if (lo == 0) {
                    const uint4 z = make_uint4(0, 0, 0, 0);
#pragma unroll
                    for (int q = 0; q < 8; q++) {
                        reinterpret_cast<uint4 *>(blk)[q] = reinterpret_cast<const uint4 *>(lb)[q];
                        reinterpret_cast<uint4 *>(lb)[q] = z;
                    }
                } else {
                    for (int q = lo; q < 64; q++) {
                        blk[q] = lb[q];
                        lb[q] = 0;
                    }
                }
            }
            lo = 0;
            k = 0;
            b = b + 1 == B ? 0 : b + 1;
            nb++;
            bi++;
            comp = G.b2comp[b];
            dct = &T.dc[comp];
            act = &T.ac[comp];
            if (WRITE && b == 0) { // next MCU
                mx++;
                if (mx == (unsigned)P.mbs_wide) {
                    mx = 0;
                    my++;
                }
            }
            live = WRITE && bi < need;
            if (live) blk = block_ptr(b);
        }
    }
    if (live) // the block in progress at the exit: this lane owns [lo, k); the next lane starts at index k
        for (int q = lo; q < k; q++) blk[q] = lb[q];
}

} // namespace

// A staged subsequence: its S / 32 dwords and one more.  A symbol starts before bit S and is at most 32 bits long
// (code <= 16, magnitude <= 16), so bit S + 31 is the last one any walk looks at.  S / 32 + 1 is odd: lanes reading
// the same dword index of their rows hit different LDS banks.
constexpr int SROW = S / 32 + 1;
static_assert((SROW & 1) == 1, "odd row stride");
__device__ __forceinline__ void stage_row(unsigned *row, const uint8_t *seg) {
    const uint4 *src = reinterpret_cast<const uint4 *>(seg);
#pragma unroll
    for (int q = 0; q < S / 128; q++) {
        const uint4 v = src[q];
        row[4 * q + 0] = __builtin_bswap32(v.x);
        row[4 * q + 1] = __builtin_bswap32(v.y);
        row[4 * q + 2] = __builtin_bswap32(v.z);
        row[4 * q + 3] = __builtin_bswap32(v.w);
    }
    row[S / 32] = __builtin_bswap32(reinterpret_cast<const unsigned *>(seg)[S / 32]); // the segment buffer has 16 bytes past every frame
}
constexpr int SPEC_T = HVC_HD_SPEC_T;

#ifdef HVC_HD_STATS
__device__ unsigned long long g_hd_stats[4];
void hd_stats_read(unsigned long long out[4]) { // experiments: read and clear
    unsigned long long z[4] = {0, 0, 0, 0};
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_hd_stats), sizeof z);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_hd_stats), z, sizeof z);
}
#endif

// 2-bit fields of a PF selmask with component 2 renamed 1 (its tables are component 1's: HdFrameTabs::flags)
__device__ __forceinline__ unsigned selmask_c2_as_c1(unsigned sel) {
    const unsigned hi = sel & 0xaaaaaaaau;
    return (sel & ~hi) | (hi >> 1);
}

// A code under a prefix that has no sub-table (HVC_HD_OVF): the canonical search of ITU-T T.81 F.2.2.3 over the lengths
// 11..16 in the table's overflow record (device memory; rare by construction -- the prefixes without a sub-table hold
// the table's least frequent symbols).  w = the next 32 bits; VAL picks the entry format.  "No such code" = entry 1.
template <bool VAL>
__device__ __noinline__ unsigned ovf_lookup(const HdOvf *o, unsigned w) {
    const unsigned w16 = w >> 16;
    for (int i = 0; i < 6; i++) {
        const unsigned d = (w16 >> (5 - i)) - (unsigned)o->mincode[i];
        if (d < (unsigned)o->count[i]) return (VAL ? o->val : o->spec)[(unsigned)o->valptr[i] + d];
    }
    return 1u;
}

template <bool RD_FREE, bool SEL1, class RD> // the lean walk of the synchronisation rounds, defined with k_hd_sync below
__device__ __forceinline__ void spec_walk(RD rd, const unsigned *row, const uint16_t *sp, const HdOvf *ovf, unsigned sel, int B, unsigned base,
                                          unsigned &p, int &k, int &b, unsigned &nb);

// One synchronisation launch (see the header comment).  Even launches write exit_a, odd ones exit_b.
// Inside the launch the 256 subsequences of a workgroup run up to INNER rounds among themselves through
// LDS (exit of lane t - 1 -> start of lane t), so a launch settles whole workgroups and the launches only
// have to carry states across workgroup boundaries.
constexpr int INNER = 24;
// MODE 0: the general walk, HdTables in LDS, bits straight from global memory.
// MODE 1 (HdParams::spec set): behind k_hd_sync this kernel follows the few chains of hand-overs that are still moving,
//   one step per inner round with one lane of a workgroup busy -- the latency of a single walk is all that counts, so
//   it is the lean one: spec_walk on the HdSpec tables, the lane's subsequence staged in LDS.
// MODE 2 (PF, per-frame tables): the same lean walk on a row staged in LDS; the tables are the frame's own record --
//   copied to LDS when every subsequence of the workgroup belongs to one frame whose tables fit two components'
//   worth (HdFrameTabs::flags), read from device memory otherwise.  (With 12 list rounds before it this kernel had
//   walks in most workgroups and the copies cost more than they saved; with 22 it has them in a few.)
template <int MODE>
__global__ __launch_bounds__(256) void k_hd_round(HdParams P, int round) {
    constexpr bool PF = MODE == 2;
    __shared__ __attribute__((aligned(16))) unsigned char Traw[MODE == 0 ? sizeof(HdTables) : sizeof(HdSpec)];
    __shared__ unsigned rows[MODE != 0 ? 256 * SROW + 2 : 1];
    HdTables &T = *reinterpret_cast<HdTables *>(Traw);
    __shared__ HdGeo G;
    __shared__ unsigned long long exits[256];
    load_geo(P, G);
    const int tid = threadIdx.x;
    const unsigned i = blockIdx.x * 256u + (unsigned)tid;
    const bool valid = i < P.total_sub;
    const unsigned f = valid ? P.frame_of[i] : 0u, j = valid ? i - P.sub_off[f] : 0u;
    const unsigned *slot = reinterpret_cast<const unsigned *>(P.ecs + P.ecs_off[f] + (size_t)j * (S / 8));
    const unsigned long long *prev = (round & 1) ? P.exit_a : P.exit_b;
    unsigned long long *cur = (round & 1) ? P.exit_b : P.exit_a;
    const unsigned base = j * (unsigned)S;
    unsigned long long st = pack_state(base, 0, 0), used = 0, ex = 0;
    bool have = false; // a decode with start `used` exists (from an earlier launch or an inner round)
    unsigned nb = 0;
    if (valid && round > 0) {
        used = P.start_used[i];
        ex = prev[i];
        nb = P.nblk[i];
        have = true;
        if (j > 0) st = prev[i - 1];
    }
    // Behind k_hd_sync almost every workgroup only finds its hand-overs in order: the 18 KB of tables are loaded when
    // a lane has to walk -- an inner round starts walks only where a neighbour's exit moved, i.e. after a walk --
    // and a launch that verifies costs the three arrays it reads, not the tables 7 000 workgroups would fetch.
    bool pf_lds = false; // MODE 2: the workgroup's one frame's tables are in LDS
    if (__syncthreads_or(valid && (!have || st != used))) { // (the barrier also publishes G)
        const unsigned *src = MODE == 1 ? reinterpret_cast<const unsigned *>(P.spec) : reinterpret_cast<const unsigned *>(P.tables);
        if (PF) {
            const unsigned i0 = blockIdx.x * 256u, i1 = min(i0 + 256u, P.total_sub) - 1u;
            const unsigned f0 = P.frame_of[i0];
            if (one_file(P, f0, P.frame_of[i1])) {
                const HdFrameTabs &ft = P.ftabs[P.tabset_of[f0]];
                pf_lds = (ft.flags & 1u) != 0u;
                src = reinterpret_cast<const unsigned *>(&ft.spec[0][0][0]);
            }
        }
        if (!PF || pf_lds) {
            unsigned *dst = reinterpret_cast<unsigned *>(Traw);
            for (unsigned q = threadIdx.x; q < sizeof(Traw) / 4; q += 256) dst[q] = src[q];
        }
        __syncthreads();
    }
    bool changed = false;
    for (int inner = 0; inner < INNER; inner++) {
        if (valid && (!have || st != used)) {
            unsigned p = (unsigned)st, err = 0;
            int k = (int)((st >> 32) & 0xffu), b = (int)((st >> 40) & 0xffu);
            nb = 0;
            if (MODE != 0) {
                unsigned *row = rows + tid * SROW; // (staged once per launch would do; a walk is 200 symbols, this is 40 instructions)
                stage_row(row, reinterpret_cast<const uint8_t *>(slot));
                auto rd = [row](unsigned q) { return row[q]; };
                const uint16_t *lt = reinterpret_cast<const uint16_t *>(Traw);
                if (MODE == 1) spec_walk<true, true>(rd, row, lt, &P.spec_ovf->o[0][0], P.slotmask, P.blocks_per_mcu, base, p, k, b, nb);
                else if (pf_lds) spec_walk<true, false>(rd, row, lt, &P.ftabs[P.tabset_of[f]].ovf[0][0], selmask_c2_as_c1(P.selmask), P.blocks_per_mcu, base, p, k, b, nb);
                else spec_walk<true, false>(rd, row, &P.ftabs[P.tabset_of[f]].spec[0][0][0], &P.ftabs[P.tabset_of[f]].ovf[0][0], P.selmask, P.blocks_per_mcu, base, p, k, b, nb);
            } else {
                walk<false>(P, G, T, slot, base, base + (unsigned)S, p, k, b, nb, 0, nullptr, err);
            }
            ex = pack_state(p, k, b);
            used = st;
            have = true;
            changed = true;
#ifdef HVC_HD_STATS // experiments: walks of the verifying launches, inner rounds they needed
            if (round > 0) {
                atomicAdd(&P.list_n[HVC_HD_LIST_N - 2], 1u);
                atomicMax(&P.list_n[HVC_HD_LIST_N - 1], (unsigned)inner + 1u);
            }
#endif
        }
        exits[tid] = ex;
        __syncthreads();
        bool again = false;
        if (valid && j > 0 && tid > 0) { // the predecessor sits in this workgroup (and in this frame, since j > 0)
            st = exits[tid - 1];
            again = st != used;
        }
        if (!__syncthreads_or(again)) break;
    }
    if (valid) {
        cur[i] = ex;
        P.start_used[i] = used;
        P.nblk[i] = nb;
        // Stamped with the launch's number, not set to 1: "did the LAST launch change anything" is then `*P.changed ==
        // last launch` with no clearing between launches (each clearing was a memset node: ~5 us of stream time apiece,
        // which a batch hides and a single file's call does not).  Benign race: every writer stores the same value.
        if (changed && round > 0) *P.changed = (unsigned)round;
    }
}

// ---------------------------------------------------------------------------
// The synchronisation rounds in their fast form.  k_hd_round spends its time in three places: its walk
// carries everything the write pass needs (90 instructions per symbol, in a loop where every branch is taken
// by some lane of the wavefront); it reads the bits dword by dword from global memory and has to wait for
// each of them (any lane's refill stalls the wavefront); and from the third round on only the few
// subsequences at the front of a not yet synchronised stretch have work, one lane here and there in wavefronts
// that take as long as ever -- eight rounds cost eight passes where three passes' worth of lanes are busy.
// k_hd_sync answers the three of them:
//   * the walk reads HdSpec entries -- bits to skip, index advance, end of block -- and nothing else;
//   * every lane's subsequence is staged in LDS (rows of 33 dwords, an odd stride: no bank conflicts), bytes
//     already swapped, and read through a 64-bit window of two registers that v_alignbit_b32 looks into:
//     no 64-bit shifts, no global loads inside the loop;
//   * the rounds are level-synchronous with WORK LISTS: round 0 walks every subsequence from the guessed
//     state, round 1 from the exit of its predecessor, and from then on round r + 1 walks exactly the
//     subsequences whose predecessor's exit CHANGED in round r (the lane that sees its exit change appends
//     its successor), packed densely into wavefronts.  Exits of round r go to exit buffer r & 1, which round
//     r + 1 only reads, so a round never sees a half-updated neighbour; the canonical copy (exit_a), the
//     start state and the block count belong to the subsequence's own lane.  ~3.2 passes per subsequence
//     instead of 8 on the reference frames.
// The rounds are enqueued back to back (SYNC_ROUNDS of them, the later ones over empty lists most of the
// time).  What they leave -- start_used / exit_a / nblk -- is what k_hd_round leaves; k_hd_round(1..) then
// verifies every hand-over and keeps going where a stream needs more rounds (smooth content with its periodic
// bit patterns can take hundreds), and the write pass compares the exit of its own walk with the recorded one.
constexpr int SYNC_ROUNDS = 12, SYNC_ROUNDS_PF = 22;
constexpr unsigned SYNC_TAIL_MAX_SUB = 32768; // up to four 1080p files of 1 MB: k_hd_sync_tail
constexpr int SYNC_TAIL_FROM = 5;

// rd(i) = dword i of the subsequence, big-endian order restored (i <= S / 32: one dword past it).  sp = the tables of
// the frame: [slot or component][DC, AC][SPEC_T] -- in LDS (one set for the whole batch) or, in PF mode, in device
// memory (this frame's record); sel = HdParams::selmask.
// The loop body is straight-line code but for the second-level look-up: in a wavefront of 64 walks SOME lane refills
// its window or ends a block in nine iterations out of ten, so a branch around either is paid every time, plus its
// mask bookkeeping -- and a refill inside a branch made the wavefront wait for its LDS read on the spot.  Selects
// instead; with RD_FREE (rows in LDS) the dword after the window is simply read again in every iteration (rd(ni) is a
// function of ni), and nothing waits for it before the next table look-up has come back anyway.
// RD_FREE: the row sits in LDS at `row` (SROW dwords and two more that may be read, whatever they hold).
// SEL1: sel has one bit per block of an MCU (HdParams::slotmask: the two slots of HdSpec); otherwise two (selmask).
// ovf = the overflow records of the same tables, [slot or component][DC, AC] (device memory).
template <bool RD_FREE, bool SEL1, class RD>
__device__ __forceinline__ void spec_walk(RD rd, const unsigned *row, const uint16_t *sp, const HdOvf *ovf, unsigned sel, int B, unsigned base,
                                          unsigned &p, int &k, int &b, unsigned &nb) {
    // The bit position is kept as mm = ~(P + 31), P = bits consumed since the start of the row (P < 32 + S at the
    // start): its low five bits are what v_alignbit has to shift {hi, lo} by, the window moves on by a dword when mm
    // changes above bit 4, and "p < limit" is "mm > ~(S + 31)" -- one subtraction per symbol keeps all of that current.
    // lo = dword (P + 31) >> 5 of the row, hi the one before (not looked at when P is a multiple of 32), nx the one
    // after -- read from *np, which moves with the window (up to two dwords past the row: nothing looks at those).
    const unsigned P0 = p - base;
    unsigned mm = ~(P0 + 31u);
    const unsigned mm_limit = ~((unsigned)S + 31u);
    const unsigned l0 = (P0 + 31u) >> 5;
    unsigned hi = l0 ? rd(l0 - 1u) : 0u;
    unsigned lo = rd(min(l0, (unsigned)(SROW - 1)));
    unsigned nx = rd(min(l0 + 1u, (unsigned)(SROW - 1)));
    const unsigned *np = row + l0 + 1u;
    auto tables_of = [&](int bb) -> const uint16_t * {
        return sp + __umul24(SEL1 ? __builtin_amdgcn_ubfe(sel, (unsigned)bb, 1u) : (sel >> (2 * bb)) & 3u, 2u * SPEC_T); // (a 32-bit multiply runs at a quarter of the rate)
    };
    const uint16_t *bt = tables_of(b);
    while (mm > mm_limit) {
        const unsigned w = __builtin_amdgcn_alignbit(hi, lo, mm); // the next 32 bits
        const uint16_t *t = bt + (k ? SPEC_T : 0);
        unsigned e = t[w >> 22];
        unsigned used = e & 63u;
        if (used == 0u) { // the code is longer than the first level's 10 bits
            const unsigned sn = e >> 6;
            if (sn != HVC_HD_OVF) e = t[1024u + sn * 64u + ((w >> 16) & 63u)];
            else e = ovf_lookup<false>(ovf + ((SEL1 ? __builtin_amdgcn_ubfe(sel, (unsigned)b, 1u) : (sel >> (2 * b)) & 3u) * 2u + (k ? 1u : 0u)), w);
            used = e & 63u;
        }
        k += (int)((e >> 6) & 127u); // an EOB advances by 64
#ifdef HVC_HD_STATS
        nb += 1u << 16; // experiments: symbols of this walk in the upper half (the caller takes it out again)
#endif
        const unsigned mn = mm - used;
        const bool refill = ((mn ^ mm) >> 5) != 0u; // the window's first dword is used up (<= 32 bits a symbol: one step is enough)
        mm = mn;
        if (RD_FREE) {
            hi = refill ? lo : hi;
            lo = refill ? nx : lo;
            np += refill ? 1 : 0;
            nx = *np;
        } else if (refill) {
            hi = lo;
            lo = nx;
            nx = rd(min((31u - mn) >> 5, (unsigned)(SROW - 1)));
        }
        // EOB, index 63 written, or past it (the model raises: the true parse never gets here)
        const bool end = k >= 64;
        const int b1 = b + 1 == B ? 0 : b + 1;
        k = end ? 0 : k;
        b = end ? b1 : b;
        nb += end ? 1u : 0u;
        bt = tables_of(b);
    }
    p = base + ~mm - 31u;
}

// 512 lanes per workgroup: 12 KB of tables + 66 KB of rows = 78 KB, two workgroups = 16 wavefronts per CU.  (With
// 256 lanes the tables weigh twice as much per lane and 12 wavefronts fit; the loop is latency-bound enough --
// 8 wavefronts per CU were 1.4x slower -- for the 16 to show.)
constexpr int SYNC_WG = 512;
// One subsequence of one round: true when its successor has to start again.  PF: the frame's own tables, in device
// memory -- or, with pf_lds (the same in every lane), in sp_lds: the tables of the one frame all of the workgroup's
// lanes are in; otherwise the batch's one set in LDS.
template <bool PF>
__device__ __forceinline__ bool sync_one(const HdParams &P, int round, bool valid, unsigned i, unsigned *row, const uint16_t *sp_lds,
                                         const unsigned long long *pe, unsigned long long *ce, bool pf_lds = false) {
    const unsigned f = P.frame_of[i], j = i - P.sub_off[f];
    const unsigned base = j * (unsigned)S;
    unsigned long long st = pack_state(base, 0, 0); // the guess; the truth for j == 0
    if (round > 0 && j > 0) st = pe[i - 1];
    if (!(valid && (round == 0 || st != P.start_used[i]))) return false;
    stage_row(row, P.ecs + P.ecs_off[f] + (size_t)j * (S / 8));
    unsigned p = (unsigned)st, nb = 0;
    int k = (int)((st >> 32) & 0xffu), b = (int)((st >> 40) & 0xffu);
    auto rd = [row](unsigned q) { return row[q]; };
    if (!PF) spec_walk<true, true>(rd, row, sp_lds, &P.spec_ovf->o[0][0], P.slotmask, P.blocks_per_mcu, base, p, k, b, nb);
    else if (pf_lds) spec_walk<true, false>(rd, row, sp_lds, &P.ftabs[P.tabset_of[f]].ovf[0][0], selmask_c2_as_c1(P.selmask), P.blocks_per_mcu, base, p, k, b, nb);
    else spec_walk<true, false>(rd, row, &P.ftabs[P.tabset_of[f]].spec[0][0][0], &P.ftabs[P.tabset_of[f]].ovf[0][0], P.selmask, P.blocks_per_mcu, base, p, k, b, nb);
#ifdef HVC_HD_STATS // experiments, round 0: symbols walked / 64 x the longest walk of each wavefront (what it costs)
    {
        const unsigned nsym = nb >> 16;
        nb &= 0xffffu;
        if (round == 0) {
            unsigned mx = nsym;
            for (int o = 32; o; o >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, o));
            atomicAdd(&g_hd_stats[0], (unsigned long long)nsym);
            if ((threadIdx.x & 63) == 0) atomicAdd(&g_hd_stats[1], 64ull * mx);
        }
    }
#endif
    const unsigned long long ex = pack_state(p, k, b);
    const bool differs = round == 0 || ex != P.exit_a[i];
    P.exit_a[i] = ex;
    ce[i] = ex;
    P.start_used[i] = st;
    P.nblk[i] = nb;
    // the successor (same frame) has to start again; after round 0 everybody does, no list needed
    return round > 0 && differs && i + 1 < P.sub_off[f + 1];
}

template <bool PF>
__global__ __launch_bounds__(SYNC_WG) void k_hd_sync(HdParams P, int round) {
    __shared__ uint16_t sp[2 * 2 * SPEC_T];
    __shared__ unsigned rows[SYNC_WG / 64][64 * SROW + 2]; // (+ 2: see spec_walk)
    const unsigned count = round < 2 ? P.total_sub : P.list_n[round];
    // the late rounds are launched over lists that hold a few hundred entries or none: a workgroup without work
    // leaves before it fetches 12 KB of tables (55 us a round for the 512 workgroups of such a launch, 10 without)
    if (blockIdx.x * (unsigned)SYNC_WG >= count) return;
    // PF: in rounds 0 and 1 a workgroup takes 512 consecutive subsequences -- nearly always of ONE frame, whose tables
    // (two components' worth: see HdFrameTabs::flags) then go to LDS like the batch-wide ones; the list rounds mix
    // frames within a wavefront and keep reading the frames' records in device memory.
    bool pf_lds = false;
    const unsigned *src = reinterpret_cast<const unsigned *>(P.spec);
    if (PF && round < 2) {
        const unsigned i0 = blockIdx.x * (unsigned)SYNC_WG, i1 = min(i0 + (unsigned)SYNC_WG, count) - 1u;
        const unsigned f0 = P.frame_of[i0];
        if (one_file(P, f0, P.frame_of[i1])) {
            const HdFrameTabs &ft = P.ftabs[P.tabset_of[f0]];
            pf_lds = (ft.flags & 1u) != 0u;
            src = reinterpret_cast<const unsigned *>(&ft.spec[0][0][0]);
        }
    }
    if (!PF || pf_lds) {
        unsigned *dst = reinterpret_cast<unsigned *>(sp);
        for (unsigned i = threadIdx.x; i < sizeof(HdSpec) / 4; i += SYNC_WG) dst[i] = src[i];
        __syncthreads(); // the tables; from here on the wavefronts have nothing to do with one another
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned *row = rows[wave] + lane * SROW;
    const unsigned *list = (round & 1) ? P.list1 : P.list0; // rounds 0 and 1: every subsequence, no list
    unsigned *next = (round & 1) ? P.list0 : P.list1;
    const unsigned long long *pe = (round & 1) ? P.exit_b : P.exit_c; // exits of round - 1
    unsigned long long *ce = (round & 1) ? P.exit_c : P.exit_b;       // exits of this round
    // One place in the next list per WORKGROUP and pass: a counter that every wavefront adds to by itself takes the
    // adds one after the other -- 18 ns each, and rounds 1 to 3 (31 000, 16 000, 8 000 wavefronts) took exactly that long.
    __shared__ unsigned wcount[2][SYNC_WG / 64], wbase[2]; // (two sets, by trip: a fast wavefront's next trip must not write what a slow one still reads)
    int trip = 0;
    for (unsigned tb = blockIdx.x * (unsigned)SYNC_WG; tb < count; tb += gridDim.x * (unsigned)SYNC_WG, trip ^= 1) { // (the same trips for every wavefront)
        const unsigned t = tb + (unsigned)threadIdx.x;
        const bool valid = t < count;
        const unsigned i = !valid ? 0u : round < 2 ? t : list[t];
        const bool push = sync_one<PF>(P, round, valid, i, row, sp, pe, ce, pf_lds);
        const unsigned long long m = __ballot(push);
        if (round == 0) continue; // (nothing is pushed: round 1 takes every subsequence)
        if (lane == 0) wcount[trip][wave] = (unsigned)__popcll(m);
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned total = 0;
            for (int q = 0; q < SYNC_WG / 64; q++) total += wcount[trip][q];
            wbase[trip] = total ? atomicAdd(&P.list_n[round + 1], total) : 0u;
        }
        __syncthreads();
        unsigned at = wbase[trip];
        for (int q = 0; q < wave; q++) at += wcount[trip][q];
        if (push) next[at + (unsigned)__popcll(m & ((1ull << lane) - 1ull))] = i + 1;
    }
}

// PF mode, the rounds per FRAME: workgroup (x, f) takes 512 entries of frame f -- its subsequences themselves in
// rounds 0 and 1, its own work list afterwards (HdParams::list_fn) -- so that every lane of a workgroup reads the same
// tables and they can sit in LDS (where the frame's third component shares the second's: HdFrameTabs::flags; otherwise
// the lanes read the frame's record in device memory as before).  With batch-wide lists the list rounds mixed frames
// within a wavefront and paid a look-up in device memory per symbol: 20 rounds of 22.
__global__ __launch_bounds__(SYNC_WG) void k_hd_sync_pf(HdParams P, int round) {
    __shared__ uint16_t sp[2 * 2 * SPEC_T];
    __shared__ unsigned rows[SYNC_WG / 64][64 * SROW + 2]; // (+ 2: see spec_walk)
    __shared__ unsigned wcount[2][SYNC_WG / 64], wbase[2];
    // one list per FILE (blockIdx.y): its frame, or -- restart intervals -- its frames, which share its tables
    const unsigned ipf = P.rst_ipf > 1u ? P.rst_ipf : 1u, g = blockIdx.y, n_groups = gridDim.y;
    const unsigned f = g * ipf, f_end = min(f + ipf, (unsigned)P.n_frames), s0 = P.sub_off[f];
    const unsigned count = round < 2 ? P.sub_off[f_end] - s0 : P.list_fn[(unsigned)round * n_groups + g];
    if (blockIdx.x * (unsigned)SYNC_WG >= count) return;
    const HdFrameTabs &ft = P.ftabs[P.tabset_of[f]];
    const bool pf_lds = (ft.flags & 1u) != 0u;
    if (pf_lds) {
        const unsigned *src = reinterpret_cast<const unsigned *>(&ft.spec[0][0][0]);
        unsigned *dst = reinterpret_cast<unsigned *>(sp);
        for (unsigned i = threadIdx.x; i < sizeof(sp) / 4; i += SYNC_WG) dst[i] = src[i];
        __syncthreads();
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned *row = rows[wave] + lane * SROW;
    const unsigned *list = ((round & 1) ? P.list1 : P.list0) + s0;
    unsigned *next = ((round & 1) ? P.list0 : P.list1) + s0;
    unsigned *next_n = P.list_fn + (unsigned)(round + 1) * n_groups + g;
    const unsigned long long *pe = (round & 1) ? P.exit_b : P.exit_c; // exits of round - 1
    unsigned long long *ce = (round & 1) ? P.exit_c : P.exit_b;       // exits of this round
    int trip = 0;
    for (unsigned tb = blockIdx.x * (unsigned)SYNC_WG; tb < count; tb += gridDim.x * (unsigned)SYNC_WG, trip ^= 1) {
        const unsigned t = tb + (unsigned)threadIdx.x;
        const bool valid = t < count;
        const unsigned i = !valid ? s0 : round < 2 ? s0 + t : list[t];
        const bool push = sync_one<true>(P, round, valid, i, row, sp, pe, ce, pf_lds);
        const unsigned long long m = __ballot(push);
        if (round == 0) continue; // (nothing is pushed: round 1 takes every subsequence)
        if (lane == 0) wcount[trip][wave] = (unsigned)__popcll(m);
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned total = 0;
            for (int q = 0; q < SYNC_WG / 64; q++) total += wcount[trip][q];
            wbase[trip] = total ? atomicAdd(next_n, total) : 0u;
        }
        __syncthreads();
        unsigned at = wbase[trip];
        for (int q = 0; q < wave; q++) at += wcount[trip][q];
        if (push) next[at + (unsigned)__popcll(m & ((1ull << lane) - 1ull))] = i + 1;
    }
}

// The same rounds for a handful of files (one at a time is how the model's decode_a_frame is called): from round 5
// on the lists hold a few hundred subsequences, and what a round costs is a launch and the latency of one walk.  One
// workgroup keeps the tables and goes from round to round by itself -- a barrier instead of a launch, the list lengths
// in LDS -- until a list is empty (or last_round: k_hd_round then continues).  A workgroup's own stores are visible to
// its own later loads (one CU, one L1), so the per-round buffers work as they do across launches.
template <bool PF>
__global__ __launch_bounds__(SYNC_WG) void k_hd_sync_tail(HdParams P, int first_round, int last_round) {
    __shared__ uint16_t sp[PF ? 2 : 2 * 2 * SPEC_T];
    __shared__ unsigned rows[SYNC_WG / 64][64 * SROW + 2]; // (+ 2: see spec_walk)
    __shared__ unsigned cnt_s[2];
    if (!PF) {
        const unsigned *src = reinterpret_cast<const unsigned *>(P.spec);
        unsigned *dst = reinterpret_cast<unsigned *>(sp);
        for (unsigned i = threadIdx.x; i < sizeof(HdSpec) / 4; i += SYNC_WG) dst[i] = src[i];
    }
    if (threadIdx.x == 0) {
        cnt_s[first_round & 1] = P.list_n[first_round];
        cnt_s[(first_round + 1) & 1] = 0;
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned *row = rows[wave] + lane * SROW;
    for (int round = first_round; round <= last_round; round++) {
        const unsigned count = cnt_s[round & 1];
        if (count == 0) break; // (the same value in every lane)
        const unsigned *list = (round & 1) ? P.list1 : P.list0;
        unsigned *next = (round & 1) ? P.list0 : P.list1;
        const unsigned long long *pe = (round & 1) ? P.exit_b : P.exit_c;
        unsigned long long *ce = (round & 1) ? P.exit_c : P.exit_b;
        for (unsigned t0 = (unsigned)wave * 64u; t0 < count; t0 += (unsigned)SYNC_WG) {
            const unsigned t = t0 + (unsigned)lane;
            const bool valid = t < count;
            const unsigned i = valid ? list[t] : 0u;
            const bool push = sync_one<PF>(P, round, valid, i, row, sp, pe, ce);
            const unsigned long long m = __ballot(push);
            if (m) {
                unsigned at = 0;
                if (lane == 0) at = atomicAdd(&cnt_s[(round + 1) & 1], (unsigned)__popcll(m));
                at = __shfl(at, 0);
                if (push) next[at + (unsigned)__popcll(m & ((1ull << lane) - 1ull))] = i + 1;
            }
        }
        __syncthreads(); // this round's stores before the next round's loads
        if (threadIdx.x == 0) cnt_s[round & 1] = 0; // the counter of round + 2
        __syncthreads();
    }
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
        if (carry_s < frame_ref(P, (unsigned)frame).need) atomicOr(P.status, 4u); // the stream ends before the frame does
    }
}

// The write pass: every lane decodes its subsequence from its (now true) start and stores coefficients.
__global__ __launch_bounds__(256) void k_hd_write(HdParams P, int final_round) {
    __shared__ HdTables T;
    __shared__ HdGeo G;
    __shared__ uint4 lbuf[256 * 8]; // 64 int16 per lane
    load_geo(P, G);
    {
        const unsigned *src = reinterpret_cast<const unsigned *>(P.tables);
        unsigned *dst = reinterpret_cast<unsigned *>(&T);
        for (unsigned i = threadIdx.x; i < sizeof(HdTables) / 4; i += 256) dst[i] = src[i];
#pragma unroll
        for (int q = 0; q < 8; q++) lbuf[threadIdx.x * 8 + q] = make_uint4(0, 0, 0, 0);
    }
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    const bool valid = i < P.total_sub;
    const unsigned f = valid ? P.frame_of[i] : 0u, j = valid ? i - P.sub_off[f] : 0u;
    const unsigned *slot = reinterpret_cast<const unsigned *>(P.ecs + P.ecs_off[f] + (size_t)j * (S / 8));
    __syncthreads();
    // Launched behind the synchronisation rounds without a look at their verdict (no host round trip): when the last
    // round still changed something the hand-overs are not consistent, block indices and states need not agree, and
    // nothing may be stored -- the caller sees the flag and redoes the chunk (k_hd_write2 and k_hd_dc likewise).
    if (hd_unsettled(P, final_round)) return;
    if (!valid) return;
    const unsigned first_block = P.nblk[i];
    const FrameRef R = frame_ref(P, f);
    if (first_block >= R.need) return; // past the last coded block: the model never reads this far
    const unsigned long long st = P.start_used[i];
    unsigned p = (unsigned)st;
    int k = (int)((st >> 32) & 0xffu), b = (int)((st >> 40) & 0xffu);
    unsigned nb = 0, err = 0;
    if ((unsigned)b != first_block % (unsigned)P.blocks_per_mcu) { // the MCU coordinates below count from first_block, the
        atomicOr(P.status, 8u);                                    // block-in-MCU index from the state: they must agree
        return;
    }
    const unsigned base = j * (unsigned)S;
    walk<true>(P, G, T, slot, base, base + (unsigned)S, p, k, b, nb, first_block, P.coefs + (size_t)R.file * P.coef_fs, err,
               reinterpret_cast<int16_t *>(lbuf + threadIdx.x * 8), R.mcu0, R.need);
    // the exit the synchronisation launches recorded for this subsequence must be the one this walk arrives at
    const unsigned long long *fin = (final_round & 1) ? P.exit_a : P.exit_b; // launch final_round - 1 wrote it
    if (pack_state(p, k, b) != fin[i]) err |= 8u;
    if (err) atomicOr(P.status, err);
}

// The write pass in its fast form (needs HdSpec's slots: at most two table sets).
//   * Subsequences staged in LDS and read through the two-register window, like k_hd_sync.
//   * A block belongs to the lane it STARTS in: that lane decodes on past the end of its subsequence (into the
//     rows of its neighbours, up to three of them: 64 symbols of at most 32 bits) until the block ends, and a lane
//     that starts in the middle of a block walks to its end without storing.  Every block leaves whole.
//   * Coefficients are assembled in a 128-byte LDS buffer per lane; finished blocks are stored by the WAVEFRONT:
//     the lanes concerned put (place in the records, lane) on a small list, then eight lanes per block move one
//     16-byte piece each from LDS to the record and clear it.
//   * Block ends are batched (WR_BATCH below): the symbols run in an inner loop that a lane leaves when it ends a
//     block, and the wavefront when WR_BATCH lanes have.
// 81.7 KB of LDS per workgroup: two workgroups per CU.
constexpr int WR_EXTRA = 3; // rows staged past the workgroup's own, for the last lanes' overrun
#ifndef HVC_WR_BATCH
#define HVC_WR_BATCH 12
#endif
constexpr int WR_BATCH = HVC_WR_BATCH; // block ends handled together

// One entry of a value table (k_hd_write2; HdFrameTabs::val), from HdTable's (length << 8) | value:
//   bits 0-4   bits the symbol takes: code + magnitude, 1..31;  0 = (first level only) the code is longer: bits 5-15
//              hold the number of the sub-table that has it
//   bits 5-8   magnitude bits (0..15)
//   bits 9-13  index advance: run of zeros + 1 (1 for a DC symbol; an EOB's does not matter);  0 marks what the model
//              raises on -- no code with this prefix (the walk steps one bit) or a DC category whose magnitude
//              decoder.ml:73-79 cannot hold; for the DC that is category 16 as well: its differences (|d| >= 32768, or
//              -32768) leave int16 or the range a JPEG DC can have, so the stream goes to the host reader either way
//              and the loop needs no range check
//   bit 15     EOB
__host__ __device__ inline unsigned val_entry(unsigned e, bool dc, bool second_level) {
    if (e & 0x8000u) return second_level ? 1u : (e & 0x7fffu) << 5; // (a pointer inside a sub-table: no such code)
    if (!e) return 1u;
    const unsigned len = e >> 8, v = e & 0xffu;
    if (dc) return v >= 16u ? len : (len + v) | (v << 5) | (1u << 9);
    const unsigned size = v & 15u, run = v >> 4;
    return (len + size) | (size << 5) | ((run + 1u) << 9) | (v ? 0u : 0x8000u);
}

// GBITS: the bits come straight from global memory (a dword per refill, requested one refill ahead) instead of rows
// staged in LDS -- 34 KB less of it per 256 lanes, which is what holds the staged form at two wavefronts per SIMD.
template <bool PF, bool GBITS, int WG>
__global__ __launch_bounds__(WG) void k_hd_write2(HdParams P, int final_round) {
    __shared__ uint16_t tv[2 * 2 * SPEC_T]; // value tables: [slot][DC, AC] -- PF: of the workgroup's one frame, if it is one (else they stay in device memory)
    __shared__ HdGeo G;
    __shared__ unsigned rows[GBITS ? 1 : (WG + WR_EXTRA + 2) * SROW]; // (+ 2: what a lane about to give up at `hard` may still read, unstaged)
    __shared__ uint4 lbuf[WG * 8]; // 64 int16 per lane
    __shared__ uint2 flist[WG / 64][64];  // per wavefront: blocks to store (offset in the batch's records in 16-byte units, lane)
    load_geo(P, G);
    {
        for (int sl = 0; sl < (PF ? 0 : 2); sl++)
            for (int cls = 0; cls < 2; cls++) {
                const HdTable *src_t = cls ? &P.tables->ac[P.slot_rep[sl]] : &P.tables->dc[P.slot_rep[sl]];
                const uint16_t *src = reinterpret_cast<const uint16_t *>(src_t);
                for (unsigned i = threadIdx.x; i < sizeof(HdTable) / 2; i += WG) tv[(sl * 2 + cls) * SPEC_T + i] = (uint16_t)val_entry(src[i], cls == 0, i >= 1024u);
            }
#pragma unroll
        for (int q = 0; q < 8; q++) lbuf[threadIdx.x * 8 + q] = make_uint4(0, 0, 0, 0);
    }
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const unsigned i = blockIdx.x * (unsigned)WG + (unsigned)tid;
    const bool valid = i < P.total_sub;
    const unsigned f = valid ? P.frame_of[i] : 0u, j = valid ? i - P.sub_off[f] : 0u;
    if (hd_unsettled(P, final_round)) return; // not settled: nothing is stored (see k_hd_write); uniform, before any barrier
    // PF: the workgroup's subsequences are consecutive -- nearly always of ONE frame, whose tables (two components'
    // worth: HdFrameTabs::flags) then go to LDS like the batch-wide ones
    bool pf_lds = false;
    if (PF) {
        const unsigned i0 = blockIdx.x * (unsigned)WG, i1 = min(i0 + (unsigned)WG, P.total_sub) - 1u;
        const unsigned f0 = P.frame_of[i0];
        if (one_file(P, f0, P.frame_of[i1])) {
            const HdFrameTabs &ft = P.ftabs[P.tabset_of[f0]];
            if (ft.flags & 1u) {
                pf_lds = true;
                const unsigned *src = reinterpret_cast<const unsigned *>(&ft.val[0][0][0]);
                unsigned *dst = reinterpret_cast<unsigned *>(tv);
                for (unsigned q = threadIdx.x; q < sizeof(tv) / 4; q += WG) dst[q] = src[q];
            }
        }
    }
    if (!GBITS) {
        if (valid) stage_row(rows + tid * SROW, P.ecs + P.ecs_off[f] + (size_t)j * (S / 8));
        if (tid < WR_EXTRA) {
            const unsigned i2 = blockIdx.x * (unsigned)WG + (unsigned)WG + (unsigned)tid;
            if (i2 < P.total_sub) {
                const unsigned f2 = P.frame_of[i2];
                stage_row(rows + (WG + tid) * SROW, P.ecs + P.ecs_off[f2] + (size_t)(i2 - P.sub_off[f2]) * (S / 8));
            }
        }
    }
    __syncthreads();
    const unsigned *row = rows + (GBITS ? 0 : tid * SROW);
    // the lane's subsequence in the segment buffer (the following ones come behind it; past the frame's last one
    // there is padding, another frame or the slack behind the buffer: nothing a block of this frame can reach)
    const unsigned *gbits = reinterpret_cast<const unsigned *>(P.ecs + P.ecs_off[f] + (size_t)j * (S / 8));
    // dword q of the stream from this lane's subsequence on.  LDS: rows are consecutive subsequences, SROW = S / 32 + 1 --
    // and dword S / 32, the one behind the subsequence IN MEMORY, comes from the lane's own row (its 33rd word), as the
    // synchronisation walks read it: behind a frame's last subsequence that is the zero overshoot, not the first word of
    // whichever frame comes next (a symbol that starts before the subsequence's end and looks past it must look at the
    // same bits in both walks, or the exits they reach differ).  Dwords further on: the following rows.
    auto rd = [&](unsigned q) -> unsigned { return GBITS ? __builtin_bswap32(gbits[q]) : row[q + (q ? (q - 1u) >> 5 : 0u)]; };
    auto rd1 = [&](unsigned q) -> unsigned { return GBITS ? __builtin_bswap32(gbits[q]) : row[q + ((q - 1u) >> 5)]; }; // q >= 1
    static_assert(S == 1024, "rd(): S / 32 dwords per row");
    // where the coefficients of a block that is not this lane's go: the lane's own entry of the wavefront's store list
    // (written before it is read in every flush, and LDS operations of a wavefront keep their order)
    int16_t *const nowhere = reinterpret_cast<int16_t *>(&flist[wave][lane]);
    int16_t *const lb = reinterpret_cast<int16_t *>(lbuf + tid * 8);
    const int B = P.blocks_per_mcu;
    const unsigned base = j * (unsigned)S, limit = base + (unsigned)S, hard = limit + (unsigned)(WR_EXTRA * S);
    unsigned bi = valid ? P.nblk[i] : 0xffffffffu; // the block this subsequence starts in
    const FrameRef R = frame_ref(P, f);
    bool act = valid && bi < R.need;   // past the last coded block: the model never reads this far
    const unsigned long long st = act ? P.start_used[i] : 0ull;
    const unsigned p0 = act ? (unsigned)st : base;
    int k = (int)((st >> 32) & 0xffu), b = (int)((st >> 40) & 0xffu);
    // live: the block in progress is this lane's (it started here) and inside the frame -- its coefficients are stored,
    // its errors count.  A lane that starts in the middle of a block (k > 0) only walks that one to its end.
    unsigned err = 0;
    if (act && (unsigned)b != bi % (unsigned)B) { // MCU coordinates count from bi, the block-in-MCU index comes from the
        err |= 8u;                                // state: a lane where they disagree must not store anything
        act = false;
    }
    bool live = act && k == 0;
    const unsigned mcu = (act ? bi : 0u) / (unsigned)B + R.mcu0;
    unsigned my = mcu / (unsigned)P.mbs_wide, mx = mcu - my * (unsigned)P.mbs_wide; // advance by counting
    // 16-byte units from P.coefs: records are multiples of 8 coefficients apart, planes of 64 (launch_hd_finish checks the range)
    const unsigned frame_unit0 = (unsigned)(((size_t)R.file * P.coef_fs) >> 3);
    auto block_no = [&](int bb) -> unsigned {
        const int comp = G.b2comp[bb];
        return frame_unit0 + (G.coef_off[comp] >> 3) +
               ((my * (unsigned)G.v[comp] + G.b2sy[bb]) * (unsigned)G.bw[comp] + (mx * (unsigned)G.h[comp] + G.b2sx[bb])) * 8u;
    };
    // The bit position is kept as mm = ~(P + 31), P = bits consumed since the start of the lane's row: its low five
    // bits are what v_alignbit has to shift {hi, lo} by, (31 - mm) >> 5 is the dword after the window, the window moves
    // on by a dword when mm changes above bit 4, and positions compare as mm's do, the other way round -- one
    // subtraction per symbol keeps all of that current.  lo = dword (P + 31) >> 5 of the stream, hi the one before
    // (not looked at when P is a multiple of 32), nx the one after.
    unsigned mm = ~(p0 - base + 31u);
    const unsigned l0 = (p0 - base + 31u) >> 5;
    unsigned hi = l0 ? rd(l0 - 1u) : 0u, lo = rd(l0), nx = rd(l0 + 1u);
    auto pos = [&]() -> unsigned { return base + ~mm - 31u; };                  // the bit position itself (rare branches)
    auto mm_of = [&](unsigned p) -> unsigned { return ~(p - base + 31u); };
    // The tables are in LDS or in device memory: the walk is written once and instantiated per address space (a
    // pointer that could be either would make every look-up a flat load).
    auto decode_and_store = [&](const uint16_t *const tvb, const HdOvf *const ovf, const unsigned selmask) {
    const uint16_t *bt = tvb + ((selmask >> (2 * b)) & 3u) * (2 * SPEC_T);
    const uint16_t *bt_ac = bt + SPEC_T;
    // the exit the synchronisation launches recorded for this subsequence must be the one this walk arrives at
    const unsigned long long fin_i = valid ? ((final_round & 1) ? P.exit_a : P.exit_b)[i] : 0ull; // launch final_round - 1 wrote it
    // the next bit position that needs a look: the end of the lane's own subsequence, then the end of what is staged
    const unsigned mm_limit = mm_of(limit), mm_hard = mm_of(hard);
    unsigned watch = mm_limit;
    // (Errors -- anything the model raises on in a lane's own block, a hand-over that does not match -- go straight to
    // P.status from the rare branches that find them: carried in registers they cost every iteration a few copies.)
    // In a wavefront some lane ends a block in almost every iteration.  A lane that ends a block therefore WAITS
    // (pending) until WR_BATCH lanes do, or nobody else can go on; then the wavefront does all of them at once.
    unsigned cur_block = live ? block_no(b) : 0u; // where the block in progress goes
#ifdef HVC_HD_STATS
    unsigned st_sym = 0, st_trips = 0;
#endif
    while (__any(act)) {
        // Symbols: every lane that has something to go on with decodes until it ends a block, and the wavefront goes on
        // until WR_BATCH lanes have ended one (or nobody is left).  An inner loop that lanes LEAVE, not an `if` around
        // the body in one loop: with three lane flags changing inside such an `if` the mask bookkeeping at its joins
        // was 48 scalar instructions per symbol -- 40 % of all the kernel issued.
        // The lane's state is in the index: below 64 it decodes; 64 = it waits at the end of a block; 65 = it has
        // stopped for good -- one comparison decides who stays in the loop, and no lane flag changes inside it.
        unsigned npend = 0; // (the same in every lane)
        if (act) for (;;) {
            // One symbol.  Straight-line code but for the second-level look-up and the (once per lane) crossing of the
            // subsequence's end: whatever a branch here guards, some lane of the 64 takes it nearly every time, and
            // the branch, its masks and -- for a refill -- the wait for an LDS read inside it came on top.
            const unsigned w = __builtin_amdgcn_alignbit(hi, lo, mm); // the next 32 bits: a whole symbol
            const uint16_t *t = k ? bt_ac : bt;
            unsigned e = t[__builtin_amdgcn_ubfe(w, 22u, 10u)];
            if ((e & 31u) == 0u) { // a longer code: its prefix's sub-table, or (no sub-table: HVC_HD_OVF) the canonical search
                const unsigned sn = e >> 5;
                if (sn != HVC_HD_OVF) e = t[1024u + sn * 64u + __builtin_amdgcn_ubfe(w, 16u, 6u)];
                else e = ovf_lookup<true>(ovf + (((selmask >> (2 * b)) & 3u) * 2u + (k ? 1u : 0u)), w);
            }
            // One path for DC and AC symbols (a DC symbol advances the index from 0 to 1): see val_entry
            const unsigned used = e & 31u, size = __builtin_amdgcn_ubfe(e, 5u, 4u), adv = __builtin_amdgcn_ubfe(e, 9u, 5u);
            const bool eob = (int16_t)e < 0;
            const bool bad = adv == 0u; // "Can't find dc / ac code" (one bit further) / DC category 16 and above (the code is skipped)
            // decoder.ml:73-79 mag': `size` bits after the code; a leading 0 bit means negative, i.e. the field minus
            // (2^size - 1).  As a signed field x that is x - full where the leading bit is 0, x - ~full where it is 1.
            const int x = __builtin_amdgcn_sbfe((int)w, 32u - used, size);
            const int full = (int)__builtin_amdgcn_ubfe(0xffffffffu, 0u, size);
            const int mag = x - (full ^ (x >> 31));
            const int kn = k + (int)adv;                // one past the index this symbol's coefficient has
            const bool wrong = bad || (kn > 64 && !eob); // ... / "coefficient index out of range"
            // (a zero written at kn - 1 -- EOB, a run of 16 -- changes nothing: the indices of a block only grow)
            *((live && !wrong) ? lb + kn - 1 : nowhere) = (int16_t)mag; // |mag| < 2^15: size <= 15
            const bool end_block = eob || kn > 63; // (what is `bad` advances by 0 and is no EOB)
            k = end_block ? 64 : kn; // (64: "waits at the end of a block"; the index itself stays below)
            const unsigned mn = mm - used;
            const bool refill = ((mn ^ mm) >> 5) != 0u;
            mm = mn;
            if (!GBITS) {
                hi = refill ? lo : hi;
                lo = refill ? nx : lo;
                nx = rd1((31u - mn) >> 5); // (a function of the position, >= 1: read again rather than branched around)
            } else if (refill) {
                hi = lo;
                lo = nx;
                nx = rd1((31u - mn) >> 5);
            }
            if ((live && wrong) || mm <= watch) { // the rare things behind one branch
                if (live && wrong) atomicOr(P.status, 1u);
                if (mm <= watch) { // p >= the position watched
                    if (watch == mm_limit) {
                        // where k_hd_sync's walk of this subsequence stopped: the one symbol that takes p across the limit
                        if (pack_state(pos(), end_block ? 0 : k, end_block ? (b + 1 == B ? 0 : b + 1) : b) != fin_i) atomicOr(P.status, 8u);
                        watch = mm_hard;
                        // a block in progress is finished here unless it is not this lane's (or nobody's: past the
                        // frame); a block that ends here: see below
                        if (!end_block && !live) k = 65;
                    } else { // cannot happen: 64 symbols of <= 32 bits end a block
                        atomicOr(P.status, 1u);
                        k = 65;
                    }
                }
            }
            npend += (unsigned)__popcll(__ballot(k == 64)); // (of the lanes still in the loop)
#ifdef HVC_HD_STATS // experiments: symbols this lane decoded (split at the end of its own subsequence) / trips of the wavefront
            st_sym += mm > mm_limit ? 1u : 0x10000u;
            st_trips += (unsigned)__popcll(__ballot(true)) ? 1u : 0u;
#endif
            if (k >= 64 || npend >= (unsigned)WR_BATCH) break;
        }
        if (k == 65) act = false;
        const bool pending = act && k == 64; // the lanes that wait at the end of a block
        // What happens at the end of a block -- store it, find the next one's place -- costs more than a symbol: the
        // wavefront does it for all the lanes that wait at one (every lane is here again).
        if (__any(pending)) {
            const bool flush = pending && live;
            const unsigned long long m = __ballot(flush);
            asm volatile("" ::: "memory"); // the int16 stores above and the 16-byte reads below meet in LDS, not in the type system
            const unsigned n = (unsigned)__popcll(m);
            if (flush) {
                flist[wave][__popcll(m & ((1ull << lane) - 1ull))] = make_uint2(cur_block, (unsigned)(wave * 64 + lane));
                P.dcd[(size_t)f * P.blocks_per_frame + bi] = lb[0]; // the DC difference once more, where k_hd_dc finds it without touching the records
            }
            for (unsigned g = 0; g < n; g += 8) {
                const unsigned en = g + ((unsigned)lane >> 3);
                if (en < n) {
                    const uint2 fe = flist[wave][en];
                    uint4 *src = lbuf + fe.y * 8u + ((unsigned)lane & 7u);
                    reinterpret_cast<uint4 *>(P.coefs)[(size_t)fe.x + ((unsigned)lane & 7u)] = *src;
                    *src = make_uint4(0, 0, 0, 0);
                }
            }
            asm volatile("" ::: "memory");
            if (pending) { // on to the next block
                k = 0;
                b = b + 1 == B ? 0 : b + 1;
                bi++;
                live = bi < R.need;
                bt = tvb + ((selmask >> (2 * b)) & 3u) * (2 * SPEC_T);
                bt_ac = bt + SPEC_T;
                if (b == 0) { // next MCU
                    mx++;
                    if (mx == (unsigned)P.mbs_wide) {
                        mx = 0;
                        my++;
                    }
                }
                if (mm <= mm_limit) act = false; // the next block starts in another lane's subsequence
                else cur_block = block_no(b);
            }
        }
    }
#ifdef HVC_HD_STATS // [2] symbols inside the lanes' own subsequences, [3] beyond them (finishing a block), and per
    {                   // wavefront 64 x the trips of its symbol loop -- counted by the lane that made the most
        unsigned mx = st_trips;
        for (int o = 32; o; o >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, o));
        atomicAdd(&g_hd_stats[2], (unsigned long long)(st_sym & 0xffffu) | ((unsigned long long)(st_sym >> 16) << 32));
        if (lane == 0) atomicAdd(&g_hd_stats[3], 64ull * mx);
    }
#endif
    };
    if (!PF) decode_and_store(tv, &P.spec_ovf->o[0][0], P.selmask);
    else if (pf_lds) decode_and_store(tv, &P.ftabs[P.tabset_of[f]].ovf[0][0], selmask_c2_as_c1(P.selmask));
    else decode_and_store(&P.ftabs[P.tabset_of[f]].val[0][0][0], &P.ftabs[P.tabset_of[f]].ovf[0][0], P.selmask);
    if (err) atomicOr(P.status, err);
}

// DC differences -> DC values (decoder.ml:143): inclusive prefix sum over the component's blocks in scan
// order, one workgroup per (component, frame).
__global__ __launch_bounds__(1024) void k_hd_dc(HdParams P, int final_round) {
    __shared__ int wsum[16];
    __shared__ int carry_s;
    const int comp = blockIdx.x, frame = blockIdx.y, lane = threadIdx.x, wave = lane >> 6, wl = lane & 63;
    if (comp >= P.n_comp || hd_unsettled(P, final_round)) return; // (not settled: the write pass stored nothing)
    const HdComp &C = P.comp[comp];
    const int hv = C.h * C.v;
    const FrameRef R = frame_ref(P, (unsigned)frame);
    const unsigned n = R.need / (unsigned)P.blocks_per_mcu * (unsigned)hv; // the component's blocks in this frame
    int16_t *rec = P.coefs + (size_t)R.file * P.coef_fs + C.coef_off;
    // k_hd_write2 leaves the differences in block order of the scan, 2 bytes each (reading them out of the records
    // costs a 128-byte line apiece: the records' size in traffic for 1/64 of their content)
    const int16_t *dcd = P.dcd ? P.dcd + (size_t)frame * P.blocks_per_frame : nullptr;
    if (lane == 0) carry_s = 0;
    __syncthreads();
    bool bad = false;
    // DC_E consecutive blocks per lane and trip (a serial prefix in registers, then one scan of the lanes' totals): a
    // 1080p luma component is 4 trips of the workgroup instead of 32 -- what a single file's call waits for (49 -> 10 us).
    constexpr int DC_E = 8;
    for (unsigned base = 0; base < n; base += 1024u * DC_E) {
        int16_t *dcp[DC_E];
        int v[DC_E];
        int run = 0;
#pragma unroll
        for (int e = 0; e < DC_E; e++) {
            const unsigned o = base + (unsigned)lane * DC_E + (unsigned)e;
            dcp[e] = nullptr;
            v[e] = 0;
            if (o < n) {
                const unsigned m = o / (unsigned)hv, r = o - m * (unsigned)hv;
                const unsigned sy = r / (unsigned)C.h, sx = r - sy * (unsigned)C.h;
                const unsigned mf = m + R.mcu0; // the MCU's number in the file's scan
                const unsigned my = mf / (unsigned)P.mbs_wide, mx = mf - my * (unsigned)P.mbs_wide;
                dcp[e] = rec + ((size_t)(my * C.v + sy) * C.bw + (size_t)(mx * C.h + sx)) * 64;
                v[e] = dcd ? dcd[(size_t)m * (unsigned)P.blocks_per_mcu + (unsigned)C.mcu_base + r] : *dcp[e];
            }
            run += v[e];
            v[e] = run; // inclusive prefix inside the lane's run
        }
        int incl = run;
#pragma unroll
        for (int s = 1; s < 64; s <<= 1) {
            const int t = __shfl_up(incl, s);
            if (wl >= s) incl += t;
        }
        if (wl == 63) wsum[wave] = incl;
        __syncthreads();
        int wbase = 0;
        for (int q = 0; q < wave; q++) wbase += wsum[q];
        const int before = carry_s + wbase + incl - run; // everything in front of this lane's run
#pragma unroll
        for (int e = 0; e < DC_E; e++) {
            if (!dcp[e]) continue;
            const int dc = before + v[e];
            if (dc < -32768 || dc > 32767) bad = true;
            if (P.dc_plane) // 2 bytes into a compact array instead of 2 bytes into a 128-byte record (a partial-line write each)
                P.dc_plane[(size_t)R.file * P.dc_fs + (size_t)((dcp[e] - (P.coefs + (size_t)R.file * P.coef_fs)) >> 6)] = (int16_t)dc;
            else
                *dcp[e] = (int16_t)dc;
        }
        __syncthreads();
        if (lane == 1023) carry_s = before + run;
        __syncthreads();
    }
    if (bad) atomicOr(P.status, 2u);
}

// One Huffman table in the two forms the fast kernels read (SPEC_T entries each: first level, then the sub-tables).
//   spec (synchronisation walk): see HdSpec
//   val  (k_hd_write2): see val_entry
static void convert_table(const HdTable &src, bool dc, uint16_t *spec, uint16_t *val) {
    auto conv = [dc](uint16_t e) -> uint16_t {
        if (e & 0x8000u) return (uint16_t)((e & 0x7fffu) << 6); // continues in a sub-table: "0 bits" and its number
        if (!e) return 1;                      // no code: one bit further, same state
        const unsigned len = e >> 8, v = e & 0xffu;
        if (dc) return (uint16_t)(v > 16 ? len : (len + v) | (1u << 6)); // category > 16: the index stays 0
        const unsigned size = v & 15u, run = v >> 4;
        if (!size && !run) return (uint16_t)(len | (64u << 6)); // EOB: the index leaves the block
        return (uint16_t)((len + size) | ((run + 1u) << 6));
    };
    const uint16_t *all = reinterpret_cast<const uint16_t *>(&src); // fast[1024] then sub[HVC_HD_SUBTABLES * 64]
    static_assert(sizeof(HdTable) == SPEC_T * sizeof(uint16_t), "HdTable = first level + sub-tables");
    for (int q = 0; q < SPEC_T; q++) {
        const uint16_t e = all[q];
        if (spec) spec[q] = (q >= 1024 && (e & 0x8000u)) ? (uint16_t)1 : conv(e);
        if (val) val[q] = (uint16_t)val_entry(e, dc, q >= 1024);
    }
}

// The overflow record of one table in the walks' form: the canonical search data as they are, and for every canonical
// position the symbol's entry in the two formats.
static void convert_ovf(const HdOvfRaw &src, bool dc, HdOvf &o) {
    std::memset(&o, 0, sizeof o);
    for (int i = 0; i < 6; i++) {
        o.mincode[i] = src.mincode[i];
        o.count[i] = src.used ? src.count[i] : (uint16_t)0; // (a table without overflow is never searched)
        o.valptr[i] = src.valptr[i];
    }
    for (int k = 0; k < 256; k++) {
        const unsigned len = src.lens[k], v = src.vals[k];
        uint16_t sp = 1, vl = 1;
        if (len) {
            const HdTable *none = nullptr;
            (void)none;
            const unsigned e = (len << 8) | v;
            // HdSpec's entry (see convert_table's conv)
            if (dc) sp = (uint16_t)(v > 16 ? len : (len + v) | (1u << 6));
            else {
                const unsigned size = v & 15u, run = v >> 4;
                sp = (!size && !run) ? (uint16_t)(len | (64u << 6)) : (uint16_t)((len + size) | ((run + 1u) << 6));
            }
            vl = (uint16_t)val_entry(e, dc, true);
        }
        o.spec[k] = sp;
        o.val[k] = vl;
    }
}

bool tables_use_overflow(const HdTables &t, int n_comp) {
    for (int c = 0; c < n_comp && c < 3; c++)
        if (t.ovf_dc[c].used || t.ovf_ac[c].used) return true;
    return false;
}

bool make_spec(const HdTables &t, int n_comp, HdSpec &out, unsigned char slot_of_comp[4], unsigned char slot_rep[2], HdSpecOvf *ovf) {
    if (n_comp < 1 || n_comp > 3) return false;
    int rep[2] = {0, -1}; // the component whose tables a slot holds
    for (int c = 0; c < 4; c++) slot_of_comp[c] = 0;
    for (int c = 1; c < n_comp; c++) {
        auto same = [&](int a) { return !std::memcmp(&t.dc[c], &t.dc[a], sizeof(HdTable)) && !std::memcmp(&t.ac[c], &t.ac[a], sizeof(HdTable)); };
        if (same(rep[0])) continue;
        if (rep[1] < 0) rep[1] = c;
        if (!same(rep[1])) return false;
        slot_of_comp[c] = 1;
    }
    slot_rep[0] = 0;
    slot_rep[1] = (unsigned char)(rep[1] < 0 ? 0 : rep[1]);
    std::memset(&out, 0, sizeof out);
    if (ovf) std::memset(ovf, 0, sizeof *ovf);
    for (int sl = 0; sl < 2; sl++) {
        if (rep[sl] < 0) continue;
        convert_table(t.dc[rep[sl]], true, out.t[sl][0], nullptr);
        convert_table(t.ac[rep[sl]], false, out.t[sl][1], nullptr);
        if (ovf) {
            convert_ovf(t.ovf_dc[rep[sl]], true, ovf->o[sl][0]);
            convert_ovf(t.ovf_ac[rep[sl]], false, ovf->o[sl][1]);
        }
    }
    return true;
}

void make_frame_tabs(const HdTables &t, int n_comp, HdFrameTabs &out) {
    out.flags = (n_comp < 3 || (!std::memcmp(&t.dc[2], &t.dc[1], sizeof(HdTable)) && !std::memcmp(&t.ac[2], &t.ac[1], sizeof(HdTable)))) ? 1u : 0u;
    out.pad[0] = out.pad[1] = out.pad[2] = 0;
    for (int c = 0; c < 3; c++) {
        const int s = c < n_comp ? c : 0; // components the frame does not have: a copy, never read
        convert_table(t.dc[s], true, out.spec[c][0], out.val[c][0]);
        convert_table(t.ac[s], false, out.spec[c][1], out.val[c][1]);
        convert_ovf(t.ovf_dc[s], true, out.ovf[c][0]);
        convert_ovf(t.ovf_ac[s], false, out.ovf[c][1]);
    }
}

// frame_of[i] = the frame subsequence i belongs to, from sub_off (one workgroup per frame).  The batch pipeline used to
// fill and upload this array per chunk: two million words written by the one thread that also feeds the copy engine.
// The same launch clears what the synchronisation launches count in -- the flags, the work lists' lengths (per batch and,
// in PF mode, per frame): the reader's first launch, so that no memset node stands between an upload and round 0.
__global__ __launch_bounds__(256) void k_hd_frame_of(HdParams P) {
    const unsigned f = blockIdx.x, end = P.sub_off[f + 1];
    unsigned *frame_of = const_cast<unsigned *>(P.frame_of);
    for (unsigned q = P.sub_off[f] + threadIdx.x; q < end; q += 256u) frame_of[q] = f;
    if (f == 0) {
        if (threadIdx.x < HVC_HD_LIST_N) P.list_n[threadIdx.x] = 0u;
        if (threadIdx.x == 0) {
            *P.changed = 0u;
            *P.status = 0u;
        }
    }
    if (P.list_fn && threadIdx.x < HVC_HD_LIST_N) P.list_fn[threadIdx.x * (unsigned)P.n_frames + f] = 0u;
}

// every reader run starts with this launch (launch_hd_round(P, 0, ...) relies on the cleared lists and flags)
hipError_t launch_hd_frame_of(const HdParams &P, hipStream_t s) {
    if (P.total_sub == 0 || P.n_frames <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_hd_frame_of, dim3((unsigned)P.n_frames), dim3(256), 0, s, P);
    return hipGetLastError();
}

template <bool PF>
static hipError_t launch_hd_round_t(const HdParams &P, int round, hipStream_t s) {
    if (round == 0 && (PF || P.spec)) { // all the fast rounds; launches 1.. of k_hd_round verify and, if need be, continue
        // (list_n / list_fn are zero: launch_hd_frame_of)
        const unsigned all = (P.total_sub + (unsigned)SYNC_WG - 1u) / (unsigned)SYNC_WG;
        if (P.total_sub <= SYNC_TAIL_MAX_SUB) { // a few files: five rounds as launches, the rest inside one workgroup
            for (int r = 0; r < SYNC_TAIL_FROM; r++) hipLaunchKernelGGL(k_hd_sync<PF>, dim3(all), dim3(SYNC_WG), 0, s, P, r);
            hipLaunchKernelGGL(k_hd_sync_tail<PF>, dim3(1), dim3(SYNC_WG), 0, s, P, SYNC_TAIL_FROM, SYNC_TAIL_FROM + 200);
            return hipGetLastError();
        }
        // (fewer list rounds for a single file's few thousand subsequences -- leaving the slow stretches to
        // k_hd_round's inner rounds earlier -- was tried: 10-25 % slower)
        // HVC_HD_SYNC_ROUNDS=n (experiments): fewer list rounds, leaving more to k_hd_round's inner rounds
        // Per-file tables are mostly optimised ones: codes without unused space fall into step more slowly (the lists
        // shrink to 0.72 of their length a round, not to 0.53), and what the list rounds leave is walked by the
        // verifying launches at the price of look-ups in device memory -- ten more rounds at 70 us each are cheaper.
        static const int rounds = [] {
            const char *v = getenv(PF ? "HVC_HD_SYNC_ROUNDS_PF" : "HVC_HD_SYNC_ROUNDS");
            const int most = PF ? SYNC_ROUNDS_PF : SYNC_ROUNDS, n = v ? atoi(v) : most;
            return n < 2 ? 2 : n > HVC_HD_LIST_N - 4 ? HVC_HD_LIST_N - 4 : n;
        }();
        if (PF && P.list_fn && P.max_frame_sub) { // per-frame lists (k_hd_sync_pf)
            const unsigned per_frame = (P.max_frame_sub + (unsigned)SYNC_WG - 1u) / (unsigned)SYNC_WG;
            for (int r = 0; r < rounds; r++) hipLaunchKernelGGL(k_hd_sync_pf, dim3(per_frame, hd_files(P)), dim3(SYNC_WG), 0, s, P, r);
            return hipGetLastError();
        }
        for (int r = 0; r < rounds; r++) {
            // the lists shrink by about half a round; a grid-stride loop takes whatever is there
            const unsigned grid = r < 2 ? all : min(all, r < 4 ? 2048u : 512u);
            hipLaunchKernelGGL(k_hd_sync<PF>, dim3(grid), dim3(SYNC_WG), 0, s, P, r);
        }
        return hipGetLastError();
    }
    if (PF) hipLaunchKernelGGL(k_hd_round<2>, dim3((P.total_sub + 255u) / 256u), dim3(256), 0, s, P, round);
    else if (P.spec) hipLaunchKernelGGL(k_hd_round<1>, dim3((P.total_sub + 255u) / 256u), dim3(256), 0, s, P, round);
    else hipLaunchKernelGGL(k_hd_round<0>, dim3((P.total_sub + 255u) / 256u), dim3(256), 0, s, P, round);
    return hipGetLastError();
}

hipError_t launch_hd_round(const HdParams &P, int round, hipStream_t s) {
    if (P.total_sub == 0) return hipSuccess;
    return P.ftabs ? launch_hd_round_t<true>(P, round, s) : launch_hd_round_t<false>(P, round, s);
}

// k_hd_write2 addresses the records in 16-byte units with 32 bits
bool hd_write2_fits(const HdParams &P) { return (unsigned long long)hd_files(P) * P.coef_fs < (1ull << 35); }

hipError_t launch_hd_finish(const HdParams &P, int rounds_done, hipStream_t s) {
    if (P.total_sub == 0) return hipSuccess;
    hipLaunchKernelGGL(k_hd_scan, dim3((unsigned)P.n_frames), dim3(1024), 0, s, P);
    HdParams Q = P;
    // HVC_WR_MODE (experiments): 0 = subsequences staged in LDS, 256 lanes; 1 = bits from global memory, 512 lanes;
    // 2 = bits from global memory, 256 lanes
    static const int wr_mode = [] { const char *v = getenv("HVC_WR_MODE"); return v ? atoi(v) : 0; }();
    auto write2 = [&](auto pf) {
        constexpr bool PFv = decltype(pf)::value;
        if (wr_mode == 1) hipLaunchKernelGGL((k_hd_write2<PFv, true, 512>), dim3((P.total_sub + 511u) / 512u), dim3(512), 0, s, P, rounds_done);
        else if (wr_mode == 2) hipLaunchKernelGGL((k_hd_write2<PFv, true, 256>), dim3((P.total_sub + 255u) / 256u), dim3(256), 0, s, P, rounds_done);
        else hipLaunchKernelGGL((k_hd_write2<PFv, false, 256>), dim3((P.total_sub + 255u) / 256u), dim3(256), 0, s, P, rounds_done);
    };
    if (P.ftabs) { // PF mode has no general write pass: the host side asks hd_write2_fits() before it chooses the mode
        if (!P.dcd || !hd_write2_fits(P)) return hipErrorInvalidValue;
        write2(std::true_type{});
    } else if (P.spec && P.dcd && hd_write2_fits(P)) {
        write2(std::false_type{});
    } else {
        Q.dcd = nullptr; // k_hd_write leaves the differences in the records only
        hipLaunchKernelGGL(k_hd_write, dim3((P.total_sub + 255u) / 256u), dim3(256), 0, s, P, rounds_done);
    }
    hipLaunchKernelGGL(k_hd_dc, dim3((unsigned)P.n_comp, (unsigned)P.n_frames), dim3(1024), 0, s, Q, rounds_done);
    return hipGetLastError();
}

} // namespace hvc
