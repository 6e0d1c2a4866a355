// hvc_entropy.cpp -- host side of the JPEG model around the GPU block stage:
//   * front end:  Decoder.Header.decode + Decoder.init geometry + the Huffman /
//     DC-prediction half of Decoder.decode_block, producing the coefficient
//     records the kernels consume            (jpeg/model/src/decoder.ml:5-140, 226-345, 362-395)
//   * back end:   Encoder.write_headers + rle + write_bits over the coefficient
//     planes the encode kernel produces      (jpeg/model/src/encoder.ml:127-193, 207-264, 371-418, 476-510)
// Plain C++ (no HIP): the sequential entropy coding stays on the host by design
// (BASELINE.json north_star); it is written for throughput (64-bit bit buffer,
// one table lookup per symbol, frames decoded in parallel by the caller's
// threads), not as a transliteration.  Behaviour follows the model, including its
// quirks: one table per DQT/DHT segment, the entropy-coded segment ends at the
// first marker (restart markers unsupported), reads past the end yield zero bits.
#include <cstdint>
#include <cstdlib>
#include <chrono>
#include <atomic>
#include <cstdio>
#include <cstring>
#include <memory>
#include <new>
#include <vector>

#include <emmintrin.h>
#include <tmmintrin.h>
#include <immintrin.h>

#include "../../include/hvc_jpeg.h"
#include "hvc_hdec.h"
#include "hvc_kernels.h"
#include "hvc_pool.h"

namespace hvc {
// Restart intervals (DRI / RSTn, ITU-T T.81 B.2.4.4 and E.2.4) -- BEYOND the model, which parses DRI and then cuts the scan at
// the first RSTn like at any marker (decoder.ml:56-59, 261-281).  Honoured only where the caller has asked for it
// (hvc_set_restart_markers, hvc_jpeg_entropy_decode_restart): the flag of the thread that reads the file.
thread_local bool tl_honour_restart = false;
} // namespace hvc

namespace {

using hvc::HVC_ZF;

// ---------------------------------------------------------------------------
// header parsing (Decoder.Header.decode, decoder.ml:24-70; Markers.*.decode, markers.ml)
struct ByteReader {
    const uint8_t *p;
    size_t n, pos = 0;
    int get8() { return pos < n ? p[pos++] : (pos++, 0); } // Bits.get past the end reads zeros
    int get16() { int a = get8(); return (a << 8) | get8(); }
};

struct HuffSpec {
    int lengths[16];
    int values[256];
    int total = 0;
};
struct DhtSeg { int tclass, id; HuffSpec spec; };
struct DqtSeg { int id; uint16_t q[64]; };

struct Header {
    int width = 0, height = 0, ncomp_frame = 0;
    int cid[4], ch[4], cv[4], ctq[4];
    int ncomp_scan = 0;
    int ssel[4], sdc[4], sac[4];
    std::vector<DqtSeg> dqt; // in file order; lookups take the LAST match like the model's cons-list
    std::vector<DhtSeg> dht;
    bool have_frame = false, have_scan = false;
    int restart_interval = 0; // DRI (markers.ml:186-197): the model parses it and never looks at it again
    size_t ecs_pos = 0; // byte position right after the SOS header
};

int parse_header(const uint8_t *data, size_t n, Header &h) {
    ByteReader r{data, n};
    for (;;) {
        // find_marker (decoder.ml:24-29): skip bytes until 0xff
        for (;;) {
            if (r.pos > n + 8) return HVC_E_BAD_JPEG;
            if (r.get8() == 0xff) break;
        }
        const int mc = r.get8();
        if (mc == 0xc0) { // SOF0, markers.ml:49-59
            (void)r.get16();
            (void)r.get8();
            h.height = r.get16();
            h.width = r.get16();
            h.ncomp_frame = r.get8();
            if (h.ncomp_frame > 4) return HVC_E_BAD_JPEG;
            for (int i = 0; i < h.ncomp_frame; i++) {
                h.cid[i] = r.get8();
                const int hv = r.get8();
                h.ch[i] = hv >> 4;
                h.cv[i] = hv & 15;
                h.ctq[i] = r.get8();
            }
            h.have_frame = true;
        } else if (mc == 0xda) { // SOS, markers.ml:111-129
            (void)r.get16();
            h.ncomp_scan = r.get8();
            if (h.ncomp_scan > 4) return HVC_E_BAD_JPEG;
            for (int i = 0; i < h.ncomp_scan; i++) {
                h.ssel[i] = r.get8();
                const int t = r.get8();
                h.sdc[i] = t >> 4;
                h.sac[i] = t & 15;
            }
            (void)r.get8();
            (void)r.get8();
            (void)r.get8();
            h.have_scan = true;
            h.ecs_pos = r.pos;
            return HVC_OK;
        } else if (mc == 0xdb) { // DQT, markers.ml:162-167: ONE table per segment
            (void)r.get16();
            const int pq = r.get8();
            DqtSeg s;
            s.id = pq & 15;
            const bool wide = (pq >> 4) != 0; // element_precision = 8 lsl (pq >> 4)
            if ((pq >> 4) > 1) return HVC_E_BAD_JPEG;
            for (int i = 0; i < 64; i++) s.q[i] = (uint16_t)(wide ? r.get16() : r.get8());
            h.dqt.push_back(s);
        } else if (mc == 0xc4) { // DHT, markers.ml:209-217: ONE table per segment
            (void)r.get16();
            const int tc = r.get8();
            DhtSeg s;
            s.tclass = tc >> 4;
            s.id = tc & 15;
            int total = 0;
            for (int i = 0; i < 16; i++) total += (s.spec.lengths[i] = r.get8());
            if (total > 256) return HVC_E_BAD_JPEG;
            for (int i = 0; i < total; i++) s.spec.values[i] = r.get8();
            s.spec.total = total;
            h.dht.push_back(s);
        } else if (mc == 0xdd) { // DRI: parsed and ignored, like the model (decoder.ml:56-59) -- unless the caller opts in
            (void)r.get16();
            h.restart_interval = r.get16();
        } else if (mc == 0xd8) { // SOI
        } else if ((mc >= 0xe0 && mc <= 0xef) || mc == 0xfe) { // APPn / COM: skip (decoder.ml:31-34)
            // Bits.show 16 then advance len*8: the length field itself is part of the skipped bytes
            const size_t at = r.pos;
            const int len = r.get16();
            r.pos = at + (size_t)len;
        } else {
            return HVC_E_UNSUPPORTED_MARKER; // decoder.ml:67
        }
    }
}

// ---------------------------------------------------------------------------
// Huffman decode tables (Tables.Specification.create_code_table, tables.ml:27-45; Tables.Lut.create :490-501)
#ifndef HVC_FAST_BITS
#define HVC_FAST_BITS 10 /* 11 (tables twice the size) and BMI2 shifts were measured on the GPU box's Zen 5 cores: +-1 %
                            (profiles/r03t_reader_variants.txt) */
#endif
struct Lut {
    static constexpr int FAST_BITS = HVC_FAST_BITS;
    int max_bits = 0;
    std::vector<uint16_t> e; // (length << 8) | data ; 0 = no code; indexed by max_bits peeked bits (Tables.Lut)
    uint16_t fast[1 << FAST_BITS]; // the same entries for codes of <= FAST_BITS bits, indexed by FAST_BITS bits:
                                   // 2 KB, L1-resident; 0 = longer code (or none): look in e
    // What the tables were built from last time, and whether that worked: a walk is reused by its thread, and the files of
    // a batch -- or of a loop over single files -- mostly carry the same DHT segments (the 64 K-entry table of a 16-bit
    // code is 128 KB to fill: most of what a small file's decode costs).
    HuffSpec built_from;
    int built = 0; // 0: nothing yet, 1: tables valid, -1: that specification is no usable code; for AC tables `+ 2`: pair / whole built too
    bool same_spec(const HuffSpec &s) const {
        return built != 0 && s.total == built_from.total && !std::memcmp(s.lengths, built_from.lengths, sizeof s.lengths) &&
               !std::memcmp(s.values, built_from.values, sizeof(int) * (size_t)s.total);
    }
    // build() for a specification that may be the one the tables already hold; ac: the one-lookup tables too
    bool build_for(const HuffSpec &s, bool ac_tables) {
        if (same_spec(s)) return built > 0;
        built_from = s;
        built = build(s) ? 1 : -1;
        if (built > 0) {
            if (ac_tables) build_pair();
            else build_whole_dc();
        }
        return built > 0;
    }
    bool build(const HuffSpec &s) {
        std::memset(fast, 0, sizeof fast);
        int maxb = 0;
        for (int i = 0; i < 16; i++)
            if (s.lengths[i]) maxb = i + 1;
        max_bits = maxb;
        e.assign((size_t)1 << maxb, 0);
        unsigned code = 0;
        int k = 0;
        for (int len = 1; len <= 16; len++) {
            for (int i = 0; i < s.lengths[len - 1]; i++, k++) {
                if (len > maxb) return false;
                const unsigned first = (code + (unsigned)i) << (maxb - len);
                const unsigned count = 1u << (maxb - len);
                if (first + count > e.size()) return false; // over-subscribed table
                for (unsigned j = 0; j < count; j++) e[first + j] = (uint16_t)((len << 8) | s.values[k]);
                if (len <= FAST_BITS) {
                    const unsigned f0 = (code + (unsigned)i) << (FAST_BITS - len), fc = 1u << (FAST_BITS - len);
                    for (unsigned j = 0; j < fc && f0 + j < (1u << FAST_BITS); j++)
                        fast[f0 + j] = (uint16_t)((len << 8) | s.values[k]);
                }
            }
            code = (code + (unsigned)s.lengths[len - 1]) << 1;
        }
        return true;
    }

    // AC tables only: code AND magnitude bits decoded by one lookup when length + size <= FAST_BITS
    // (most coefficients of real streams).  One 32-bit entry, one load: bits 0-7 the bits consumed (0: not covered, take
    // the two-step path), bits 8-15 how far the coefficient index moves (run + 1; WHOLE_EOB for an end of block, so that
    // "index >= 64" is the one test that ends a block), bits 16-31 the coefficient.
    static constexpr unsigned WHOLE_EOB = 128;
    uint32_t whole[1 << FAST_BITS];
    void build_whole() {
        for (unsigned w = 0; w < (1u << FAST_BITS); w++) {
            uint32_t o = 0;
            const unsigned e = fast[w];
            if (e) {
                const int len = (int)(e >> 8), run = (int)((e >> 4) & 15), size = (int)(e & 15);
                if (len + size <= FAST_BITS) {
                    int mag = 0;
                    if (size) {
                        const unsigned code = (w >> (FAST_BITS - len - size)) & ((1u << size) - 1u);
                        mag = (code & (1u << (size - 1))) ? (int)code : (int)code - (int)((1u << size) - 1); // mag'
                    }
                    const unsigned step = (mag == 0 && run == 0) ? WHOLE_EOB : (unsigned)run + 1u; // decoder.ml:131-132
                    o = (uint32_t)(len + size) | step << 8 | (uint32_t)(uint16_t)(int16_t)mag << 16;
                }
            }
            whole[w] = o;
        }
    }
    // ... and the table the hot loop reads: ONE OR TWO symbols per lookup.  Where a second whole symbol (code + magnitude)
    // follows the first inside the FAST_BITS index, the entry carries both -- the dependency chain of the walk (shift,
    // load, shift) is then paid once for two symbols.  64-bit entry: byte 0 the bits consumed by all of it (0: the first
    // symbol is not covered), byte 1 the index step of the first symbol, byte 2 the step of both together (= byte 1
    // when there is no second), byte 3 the bits of the first symbol alone (to step back when the first one completed the block and the
    // "second" is the next block's), bits 32-47 / 48-63 the two coefficients (the second = the first when there is none:
    // the loop always stores twice, the second time over the first).  An end of block is never followed by a second symbol.
    uint64_t pair[1 << FAST_BITS];
    void build_pair() {
        build_whole();
        for (unsigned w = 0; w < (1u << FAST_BITS); w++) {
            const uint32_t a = whole[w];
            uint64_t o = 0;
            if (a & 0xffu) {
                const unsigned b1 = a & 0xffu, s1 = (a >> 8) & 0xffu, v1 = a >> 16;
                unsigned btot = b1, s2 = 0, v2 = v1;
                if (s1 != WHOLE_EOB && b1 < (unsigned)FAST_BITS) {
                    const unsigned rem = (unsigned)FAST_BITS - b1;
                    const uint32_t b = whole[(w << b1) & ((1u << FAST_BITS) - 1u)]; // (the bits behind the window read as zeros:
                    if ((b & 0xffu) && (b & 0xffu) <= rem) {                        //  a symbol that needs any of them is too long)
                        btot = b1 + (b & 0xffu);
                        s2 = (b >> 8) & 0xffu;
                        v2 = b >> 16;
                    }
                }
                o = (uint64_t)btot | (uint64_t)s1 << 8 | (uint64_t)(s1 + s2) << 16 | (uint64_t)b1 << 24 | (uint64_t)(v1 & 0xffffu) << 32 |
                    (uint64_t)(v2 & 0xffffu) << 48;
            }
            pair[w] = o;
        }
    }
    // DC tables: the same for a DC symbol -- bits 0-7 the bits consumed (code + magnitude; 0: not covered), bits 16-31
    // the DC difference (mag, decoder.ml:73-96) -- when the two fit FAST_BITS; differences of 7+ bits take the two steps
    void build_whole_dc() {
        for (unsigned w = 0; w < (1u << FAST_BITS); w++) {
            uint32_t o = 0;
            const unsigned e = fast[w];
            if (e) {
                const int len = (int)(e >> 8), cat = (int)(e & 0xff);
                if (len + cat <= FAST_BITS) {
                    int mag = 0;
                    if (cat) {
                        const unsigned code = (w >> (FAST_BITS - len - cat)) & ((1u << cat) - 1u);
                        mag = (code & (1u << (cat - 1))) ? (int)code : (int)code - (int)((1u << cat) - 1);
                    }
                    o = (uint32_t)(len + cat) | (uint32_t)(uint16_t)(int16_t)mag << 16;
                }
            }
            whole[w] = o;
        }
    }
};

// ---------------------------------------------------------------------------
// bit reader over the extracted entropy-coded segment (stuffing already removed)
struct BitReader {
    const uint8_t *p; // the segment, followed by >= PAD readable zero bytes
    size_t n;         // its length in bytes
    size_t pos = 0;   // bytes loaded so far
    uint64_t buf = 0; // the next bits, MSB-aligned
    int cnt = 0;      // how many of them are valid
    // refill() tops the window up to 57..64 valid bits: enough for several symbols (a symbol of the one-lookup path is
    // <= 10 bits, any symbol <= 16 + 15), so the hot loop refills once per GROUP of symbols and a symbol costs
    // `buf <<= bits; cnt -= bits`.  The load address depends on nothing the symbols compute, so the load is off the
    // symbols' dependency chain -- only the `or` is on it.  There is no end test: past the end the stream reads as zero
    // bits (bitstream_reader.ml:19-22), the padding is zero, and hold() at every block start keeps the load position
    // inside it -- a block consumes at most 48 + 63 * 31 bits = 251 bytes.
    static constexpr size_t PAD = 288;
    inline void refill() {
        uint64_t w;
        std::memcpy(&w, p + pos, 8);
        buf |= __builtin_bswap64(w) >> cnt;
        const int adv = (63 - cnt) >> 3;
        pos += (size_t)adv;
        cnt += adv * 8;
    }
    inline void hold() {
        if (pos > n + 8) pos = n + 8; // (everything in the window is padding by then)
    }
};

// decoder.ml:73-79 mag'
inline int extend(int cat, unsigned code) {
    return (code & (1u << (cat - 1))) ? (int)code : (int)code - (int)((1u << cat) - 1);
}
// ... for the categories no JPEG has but a DHT may name (decoder.ml:81-96 reads `cat` bits for whatever the table says): up
// to 62 bits everything in mag' is defined -- `1 lsl (cat - 1)`, `-1 lsl cat` stay below OCaml's Sys.int_size = 63.
inline long long extend_wide(int cat, unsigned long long code) {
    return ((code >> (cat - 1)) & 1ull) ? (long long)code : (long long)(code - ((1ull << cat) - 1ull));
}
// OCaml's int is 63 bits wide and wraps without a word (decoder.ml:143 `coefs.(0) + dc_pred`): the sum modulo 2^64,
// read as a 63-bit two's complement number.
inline long long add63(long long a, long long b) {
    return (long long)(((unsigned long long)a + (unsigned long long)b) << 1) >> 1;
}
const int HVC_MAX_DC_CAT = 62; // beyond: OCaml leaves the shifts of mag' unspecified (a count >= Sys.int_size); refused

} // namespace

// ===========================================================================
extern "C" {

// Decoder.Header.decode + the geometry of Decoder.init (decoder.ml:294-345)
int hvc_jpeg_read_header(const uint8_t *data, size_t n, hvc_jpeg_info *info) try {
    if (!data || !info) return HVC_E_INVALID_ARG;
    std::memset(info, 0, sizeof *info);
    Header h;
    int r = parse_header(data, n, h);
    if (r) return r;
    if (!h.have_frame || !h.have_scan) return HVC_E_BAD_JPEG; // decoder.ml:283-292
    int max_h = 0, max_v = 0;
    for (int i = 0; i < h.ncomp_frame; i++) {
        if (h.ch[i] > max_h) max_h = h.ch[i];
        if (h.cv[i] > max_v) max_v = h.cv[i];
    }
    if (max_h == 0 || max_v == 0 || h.ncomp_scan == 0) return HVC_E_BAD_JPEG;
    const long rw = ((long)h.width + max_h * 8 - 1) / (max_h * 8) * (max_h * 8);
    const long rh = ((long)h.height + max_v * 8 - 1) / (max_v * 8) * (max_v * 8);
    info->width = h.width;
    info->height = h.height;
    info->n_comp = h.ncomp_scan;
    info->ecs_offset = h.ecs_pos;
    size_t coef_off = 0, pix_off = 0;
    for (int i = 0; i < h.ncomp_scan; i++) {
        int f = -1; // find_component (decoder.ml:226-230): first match
        for (int k = 0; k < h.ncomp_frame; k++)
            if (h.cid[k] == h.ssel[i]) { f = k; break; }
        if (f < 0) return HVC_E_BAD_JPEG;
        hvc_jpeg_component &c = info->comp[i];
        c.identifier = h.cid[f];
        c.hscale = h.ch[f];
        c.vscale = h.cv[f];
        c.decoded_width = (int)(rw * h.ch[f] / max_h);
        c.decoded_height = (int)(rh * h.cv[f] / max_v);
        c.actual_width = h.width * h.ch[f] / max_h;
        c.actual_height = h.height * h.cv[f] / max_v;
        // (multiples of 8 by construction.  ZERO where the component's sampling factor is zero, or the frame's width or
        // height is: Decoder.init builds an empty plane there -- Plane.create ~width:0 -- and decode_seq walks the MCUs with
        // no block for it, decoder.ml:304-345, 362-395; what cannot be made of such planes is a Frame.t: hvc_jpeg_get_yuv_frame)
        if (c.decoded_width < 0 || c.decoded_height < 0 || (c.decoded_width & 7) || (c.decoded_height & 7)) return HVC_E_BAD_JPEG;
        // find_quant_table (decoder.ml:232-236): newest table with that id
        int qi = -1;
        for (int k = (int)h.dqt.size() - 1; k >= 0; k--)
            if (h.dqt[k].id == h.ctq[f]) { qi = k; break; }
        if (qi < 0) return HVC_E_BAD_JPEG;
        int slot = -1; // de-duplicate into at most 4 table slots
        for (int k = 0; k < info->n_qtabs; k++)
            if (!std::memcmp(info->qtabs[k], h.dqt[qi].q, sizeof info->qtabs[k])) slot = k;
        if (slot < 0) {
            if (info->n_qtabs == 4) return HVC_E_BAD_JPEG;
            slot = info->n_qtabs++;
            std::memcpy(info->qtabs[slot], h.dqt[qi].q, sizeof info->qtabs[slot]);
        }
        c.dc_table = h.sdc[i];
        c.ac_table = h.sac[i];
        hvc_component &L = info->layout[i];
        L.blocks_w = c.decoded_width / 8;
        L.blocks_h = c.decoded_height / 8;
        L.qtab = slot;
        L.coef_offset = coef_off;
        L.plane_offset = pix_off;
        L.stride = (size_t)c.decoded_width;
        coef_off += (size_t)L.blocks_w * L.blocks_h * 64;
        pix_off += (size_t)c.decoded_width * c.decoded_height;
    }
    info->coef_count = coef_off;
    info->pixel_bytes = pix_off;
    return HVC_OK;
} HVC_ABI_CATCH

// The Huffman half of Decoder.decode (decode_seq order, decoder.ml:362-395; huffman_decode :118-140;
// the DC predictor add of :143) into one frame's coefficient record: int16, zig-zag order, DC absolute,
// block (bx,by) of component i at coefs + layout[i].coef_offset + (by*blocks_w + bx)*64.
//
// One file's reader as a resumable walk: step() decodes ONE symbol of the block in progress, or begins the next block
// (position check, clear_block, the DC symbol and its predictor).  A file is one stream and its symbols form one
// dependency chain -- shift, table load, shift, or: ~8 cycles a symbol with nothing else for the core to do -- but
// two FILES are two chains: entropy_decode_two() below steps two walks alternately and the out-of-order core overlaps
// them (what the batch pipelines' workers do: 515 -> 750+ Mpixel/s per thread).
namespace hvc {
size_t extract_ecs_to(const uint8_t *data, size_t n, size_t pos, uint8_t *dst, size_t cap);
}
namespace {
// The entropy-coded segment THROUGH its RSTn markers: every interval unstuffed and appended with `pad` zero bytes behind it
// (past its end an interval reads as zero bits, like a segment past its end: bitstream_reader.ml:19-22 -- and what the GPU
// reader's intervals see behind theirs); starts[k] / lens[k] = where interval k begins in `out` and how long it is; ends at
// the first marker that is not RSTn (0xFF fill bytes in front of a marker are skipped, T.81 B.1.1.2).
void extract_ecs_restart(const uint8_t *data, size_t n, size_t pos, std::vector<uint8_t> &out, std::vector<size_t> &starts,
                         std::vector<size_t> &lens, size_t pad) {
    out.clear();
    starts.clear();
    lens.clear();
    starts.push_back(0);
    out.reserve((n > pos ? n - pos : 0) + pad);
    auto close = [&]() {
        lens.push_back(out.size() - starts.back());
        out.resize(out.size() + pad, 0);
    };
    while (pos < n) {
        const uint8_t *ff = (const uint8_t *)std::memchr(data + pos, 0xff, n - pos);
        const size_t stop = ff ? (size_t)(ff - data) : n;
        out.insert(out.end(), data + pos, data + stop);
        if (!ff) break;
        const int next = stop + 1 < n ? data[stop + 1] : -1;
        if (next == 0x00) {
            out.push_back(0xff);
            pos = stop + 2;
        } else if (next == 0xff) {
            pos = stop + 1; // a fill byte
        } else if (next >= 0xd0 && next <= 0xd7) {
            close();
            starts.push_back(out.size());
            pos = stop + 2;
        } else {
            break;
        }
    }
    close();
    starts.push_back(out.size() - pad); // (a stream with fewer markers than its DRI promises: intervals of no bytes = zeros)
    lens.push_back(0);
}

// An hvc_jpeg_info is the caller's: hvc_jpeg_read_header / hvc_jpeg_encoder_layout filled it in, normally -- but nothing
// keeps a caller from changing it, and the readers and coders index and divide by what it says.  What they rely on:
// one to four components, sampling factors 1..15, planes of at least one block, and every component's record inside
// the coef_count elements the caller's buffer is said to have.  The readers (allow_empty) also take what the model's
// decoder takes: a sampling factor of zero and a plane without blocks (Decoder.init, decoder.ml:304-345).
bool info_is_sane(const hvc_jpeg_info *info, bool allow_empty = false) {
    if (info->n_comp < 1 || info->n_comp > 4) return false;
    const int lo = allow_empty ? 0 : 1;
    for (int i = 0; i < info->n_comp; i++) {
        const hvc_jpeg_component &c = info->comp[i];
        const hvc_component &L = info->layout[i];
        if (c.hscale < lo || c.hscale > 15 || c.vscale < lo || c.vscale > 15) return false;
        if (L.blocks_w < lo || L.blocks_h < lo || L.blocks_w > (1 << 20) || L.blocks_h > (1 << 20)) return false;
        if (c.decoded_width < 0 || c.decoded_height < 0) return false;
        const unsigned long long n = (unsigned long long)L.blocks_w * (unsigned long long)L.blocks_h * 64ull;
        if (L.coef_offset > info->coef_count || n > info->coef_count - L.coef_offset) return false;
    }
    return true;
}

// a finished block leaves for the record and its buffer is cleared: two 64-byte loads, streaming stores and stores
// where the CPU has them (six memory operations instead of twenty-four; the record's blocks are 128 bytes apart, so a
// 64-byte aligned record makes every block's two halves whole cache lines)
__attribute__((target("avx512f"))) static void flush_512(int16_t *dst, int16_t *cur) {
    const __m512i zero = _mm512_setzero_si512();
    const __m512i a = _mm512_load_si512(cur), b = _mm512_load_si512(cur + 32);
    _mm512_stream_si512((__m512i *)dst, a);
    _mm512_stream_si512((__m512i *)(dst + 32), b);
    _mm512_store_si512(cur, zero);
    _mm512_store_si512(cur + 32, zero);
}
struct Walk {
    const hvc_jpeg_info *info = nullptr;
    int16_t *coefs = nullptr;
    std::vector<hvc::WideDc> *wide = nullptr; // nullptr: an absolute DC outside int16 is HVC_E_RANGE; otherwise the record
                                              // gets the saturated value and the block goes on the list with its true DC
    Lut dc_tab[4], ac_tab[4];
    const Lut *dc[4] = {nullptr, nullptr, nullptr, nullptr}, *ac[4] = {nullptr, nullptr, nullptr, nullptr}; // per component:
                                              // components that name the same DHT segment share one table (Cb and Cr
                                              // do in every file an encoder writes: 12 KB less for the L1 to hold)
    std::vector<uint8_t> ecs;
    // restart intervals, where the caller has opted in (else rst_interval = 0 and nothing below is looked at): MCUs per
    // interval, where each interval's bytes begin in `ecs`, the interval in progress, MCUs begun, the MCU the next one starts at
    int rst_interval = 0;
    std::vector<size_t> rst_start, rst_len;
    size_t rst_k = 0;
    long long mcus_begun = 0, rst_at = 0;
    BitReader br{nullptr, 0};
    long long dc_pred[4] = {0, 0, 0, 0}; // (the model's 63-bit ints: 67 M blocks of +-65535 stay far inside)
    int mbs_wide = 0, mbs_high = 0;
    // The NEXT block, decode_seq order (decoder.ml:362-395: sx, sy inside the component's part of the MCU, then the
    // components, then the MCUs): block `bi` of MCU (my, mx).  The blocks of one MCU are listed once (component, place
    // inside the component's part, offset from the part's first block); per component the offset of that first block
    // moves along with (my, mx).
    struct McuBlock {
        uint8_t comp, dx, dy;
        uint32_t off; // int16 elements from the first block of the component's part of the MCU
    };
    std::vector<McuBlock> mcu;
    size_t part[4] = {0, 0, 0, 0}; // per component: int16 elements from `coefs` to block (my * vscale, mx * hscale)
    bool regular = true;            // every block of every MCU lies inside its component's planes: no per-block test
    int my = 0, mx = 0, bi = 0;
    // The block in progress is assembled HERE (one address for the whole file: L1-resident) and leaves for the record in
    // one piece when the next block begins -- eight 16-byte streaming stores where the record is 16-byte aligned: the
    // record (6 MB a frame) is written once and never read by this thread, so its lines need not be fetched for ownership
    // nor kept in the cache; clear_block (decoder.ml:109-116) is the re-zeroing of these 128 bytes.
    // (behind its 64 coefficients: room for the stores of symbols that step past the block -- an end of block steps 128,
    // an index out of range up to 15 + 16 -- which the loop performs before it looks at the index)
    alignas(64) int16_t cur[64 + 2 * 128 + 64] = {};
    int16_t *const blk = cur;
    int16_t *dst = nullptr;     // where the block in progress belongs, or null: none
    int stream_out = 0;         // 0: memcpy (the record is not 16-byte aligned), 1: 16-byte streaming stores, 2: 64-byte ones
    int k = 64;                 // index of the next coefficient of the block in progress; 64 = none in progress
    const uint64_t *acw = nullptr;
    const uint16_t *acf = nullptr, *act = nullptr;
    int amax = 0;
    bool done = false;

    int prepare(const uint8_t *data, size_t n, const hvc_jpeg_info *info_, int16_t *coefs_, std::vector<hvc::WideDc> *wide_) {
        if (!data || !info_ || !info_is_sane(info_, true)) return HVC_E_INVALID_ARG;
        if (!coefs_ && info_->coef_count) return HVC_E_INVALID_ARG; // (a record without a block needs no memory)
        // (a walk is reused by its thread: everything a previous file left behind starts over)
        for (int i = 0; i < 4; i++) dc_pred[i] = 0, dc[i] = ac[i] = nullptr, part[i] = 0;
        my = mx = bi = 0;
        k = 64;
        done = false;
        dst = nullptr;
        regular = true;
        mcu.clear();
        std::memset(cur, 0, 64 * sizeof(int16_t));
        info = info_;
        coefs = coefs_;
        static const bool have512 = __builtin_cpu_supports("avx512f") && !(std::getenv("HVC_NO_AVX512") && std::getenv("HVC_NO_AVX512")[0] == '1'); // (the switch: A/B)
        // how the blocks may leave: every block sits at coefs + a component's offset + a multiple of 128 bytes, so the
        // record's address and the offsets (the caller's: not necessarily hvc_jpeg_read_header's) decide the alignment
        uintptr_t align_bits = (uintptr_t)coefs_;
        for (int i = 0; i < info_->n_comp && i < 4; i++) align_bits |= (uintptr_t)info_->layout[i].coef_offset * sizeof(int16_t);
        stream_out = (align_bits & 63) == 0 && have512 ? 2 : (align_bits & 15) == 0 ? 1 : 0;
        wide = wide_;
        Header h;
        int r = parse_header(data, n, h);
        if (r) return r;
        int dseg[4], aseg[4];
        for (int i = 0; i < info->n_comp; i++) {
            int di = -1, ai = -1; // find_huffman_table (decoder.ml:238-259): newest match
            for (int q = (int)h.dht.size() - 1; q >= 0; q--) {
                if (di < 0 && h.dht[q].tclass == 0 && h.dht[q].id == info->comp[i].dc_table) di = q;
                if (ai < 0 && h.dht[q].tclass == 1 && h.dht[q].id == info->comp[i].ac_table) ai = q;
            }
            if (di < 0 || ai < 0) return HVC_E_BAD_JPEG;
            dseg[i] = di;
            aseg[i] = ai;
            dc[i] = ac[i] = nullptr;
            for (int j = 0; j < i; j++) {
                if (dseg[j] == di) dc[i] = dc[j];
                if (aseg[j] == ai) ac[i] = ac[j];
            }
            if (!dc[i]) {
                if (!dc_tab[i].build_for(h.dht[di].spec, false)) return HVC_E_BAD_JPEG;
                dc[i] = &dc_tab[i];
            }
            if (!ac[i]) {
                if (!ac_tab[i].build_for(h.dht[ai].spec, true)) return HVC_E_BAD_JPEG;
                ac[i] = &ac_tab[i];
            }
        }
        // extract_entropy_coded_bits (decoder.ml:261-281): up to the first marker, 0xff00 -> 0xff.  (A missing EOI just
        // ends the segment: the model would spin on zero bytes there.)
        const size_t pos = h.ecs_pos, room = (n > pos ? n - pos : 0) + 16 + BitReader::PAD;
        rst_interval = hvc::tl_honour_restart ? h.restart_interval : 0;
        {   // a scan of one interval (MCUs <= DRI) holds no RSTn: it is read as the plain segment, the same cutter
            // (extract_ecs_to) the GPU reader's rst_multi == false path takes -- so a fall-back from one reader to the other
            // never changes the records (0xFF 0xFF and a trailing 0xFF were cut differently by the two cutters)
            const hvc_jpeg_component &f = info->comp[0];
            if (rst_interval && f.hscale && f.vscale) {
                const unsigned long long mcus = (unsigned long long)(f.decoded_width / (8 * f.hscale)) * (f.decoded_height / (8 * f.vscale));
                if (mcus <= (unsigned long long)rst_interval) rst_interval = 0;
            }
        }
        rst_k = 0;
        mcus_begun = 0;
        rst_at = rst_interval;
        if (rst_interval) { // (opt-in, beyond the model)
            extract_ecs_restart(data, n, pos, ecs, rst_start, rst_len, 16 + BitReader::PAD);
            br = BitReader{ecs.data(), rst_len[0]};
        } else {
        ecs.resize(room);
        size_t got = hvc::extract_ecs_to(data, n, pos, ecs.data(), room - BitReader::PAD);
        if (got == SIZE_MAX) return HVC_E_BAD_JPEG; // (cannot happen: unstuffing only ever shortens)
        std::memset(ecs.data() + got, 0, room - got); // zero padding: see BitReader
        br = BitReader{ecs.data(), got};
        }
        const hvc_jpeg_component &c0 = info->comp[0];
        // decode_seq divides by the FIRST component's factors (decoder.ml:377-382): Division_by_zero there
        if (c0.hscale == 0 || c0.vscale == 0) return HVC_E_BAD_JPEG;
        mbs_wide = c0.decoded_width / (8 * c0.hscale);
        mbs_high = c0.decoded_height / (8 * c0.vscale);
        done = mbs_wide <= 0 || mbs_high <= 0 || info->n_comp <= 0;
        if (done) return HVC_OK;
        try {
            for (int i = 0; i < info->n_comp; i++) {
                const hvc_jpeg_component &c = info->comp[i];
                const hvc_component &L = info->layout[i];
                if ((long long)mbs_wide * c.hscale > L.blocks_w || (long long)mbs_high * c.vscale > L.blocks_h) regular = false;
                for (int y = 0; y < c.vscale; y++)
                    for (int x = 0; x < c.hscale; x++)
                        mcu.push_back(McuBlock{(uint8_t)i, (uint8_t)x, (uint8_t)y, (uint32_t)(((size_t)y * L.blocks_w + x) * 64)});
            }
        } catch (const std::bad_alloc &) {
            return HVC_E_OUT_OF_MEMORY;
        }
        if (mcu.empty()) { // (components without blocks: nothing to read)
            done = true;
            return HVC_OK;
        }
        row_parts();
        return HVC_OK;
    }

    void row_parts() { // MCU (my, 0)
        for (int i = 0; i < info->n_comp; i++)
            part[i] = info->layout[i].coef_offset + (size_t)my * info->comp[i].vscale * info->layout[i].blocks_w * 64;
    }

    // block bi of MCU (my, mx): bounds, clear_block (decoder.ml:109-116, right before the block is written: one pass
    // over the record instead of a 6 MB memset that has left the cache by the time the block comes up), the DC symbol;
    // then the position moves on
    int begin_block() {
        if (rst_interval && bi == 0 && mcus_begun == rst_at) { // the first block of an MCU that opens a restart interval:
            rst_at += rst_interval;                          // byte-aligned data behind the RSTn marker, predictors at zero (T.81 E.2.4)
            rst_k++;
            const size_t last = rst_start.size() - 1; // (a stream with fewer markers than its DRI promises: zeros from its end on)
            const size_t q = (size_t)rst_k < last ? (size_t)rst_k : last;
            br = BitReader{ecs.data() + rst_start[q], rst_len[q]};
            for (int j = 0; j < 4; j++) dc_pred[j] = 0;
        }
        const McuBlock mb = mcu[(size_t)bi];
        const int i = mb.comp;
        if (!regular) {
            const hvc_jpeg_component &c = info->comp[i];
            const hvc_component &L = info->layout[i];
            if (mx * c.hscale + mb.dx >= L.blocks_w || my * c.vscale + mb.dy >= L.blocks_h) return HVC_E_BAD_JPEG; // Plane.set out of bounds
        }
        flush();
        dst = coefs + part[i] + mb.off;
        br.hold();
        br.refill();
        long long diff = 0;
        const uint32_t dw = dc[i]->whole[br.buf >> (64 - Lut::FAST_BITS)];
        if (dw & 0xffu) { // code and magnitude in one lookup
            br.buf <<= dw & 0xffu;
            br.cnt -= (int)(dw & 0xffu);
            diff = (int16_t)(dw >> 16);
        } else if (int r = dc_two_steps(i, diff)) {
            return r;
        }
        return finish_begin(i, diff);
    }

    int dc_two_steps(int i, long long &diff) {
        unsigned e = dc[i]->fast[br.buf >> (64 - Lut::FAST_BITS)];
        if (!e) e = dc[i]->e[dc[i]->max_bits ? br.buf >> (64 - dc[i]->max_bits) : 0];
        if (!e) return HVC_E_BAD_JPEG; // "Can't find dc code"
        br.buf <<= e >> 8;
        br.cnt -= (int)(e >> 8);
        const int cat = e & 0xff;
        if (cat > 16) {
            // No JPEG has DC categories above 11 (baseline) / 16; the model, though, reads `cat`
            // magnitude bits for whatever the table says (decoder.ml:81-96: no check).  Up to 62 bits
            // this reader follows it -- such a difference never fits the int16 record (|d| >= 65536),
            // so only the wide-DC mode goes on -- beyond that OCaml itself leaves the result open.
            if (cat > HVC_MAX_DC_CAT) return HVC_E_BAD_JPEG;
            if (!wide) return HVC_E_RANGE;
            unsigned long long code = 0;
            for (int left = cat; left > 0;) { // (the window holds 32 bits and more after a refill)
                const int take = left > 32 ? 32 : left;
                br.refill();
                code = (code << take) | (br.buf >> (64 - take));
                br.buf <<= take;
                br.cnt -= take;
                left -= take;
            }
            diff = extend_wide(cat, code);
        } else if (cat) { // code + magnitude <= 32 bits: still inside the window
            diff = extend(cat, (unsigned)(br.buf >> (64 - cat)));
            br.buf <<= cat;
            br.cnt -= cat;
        }
        return HVC_OK;
    }

    inline int finish_begin(int i, long long diff) {
        const long long dcv = add63(diff, dc_pred[i]);
        dc_pred[i] = dcv;
        if (dcv < -32768 || dcv > 32767) {
            if (!wide) return HVC_E_RANGE;
            try {
                wide->push_back(hvc::WideDc{(uint32_t)((size_t)(dst - coefs) >> 6), dcv});
            } catch (const std::bad_alloc &) {
                return HVC_E_OUT_OF_MEMORY;
            }
            blk[0] = (int16_t)(dcv < 0 ? -32767 : 32767);
        } else {
            blk[0] = (int16_t)dcv;
        }
        k = 1;
        acw = ac[i]->pair;
        acf = ac[i]->fast;
        act = ac[i]->e.data();
        amax = ac[i]->max_bits;
        if (++bi == (int)mcu.size()) {
            bi = 0;
            ++mcus_begun;
            if (++mx == mbs_wide) {
                mx = 0;
                ++my;
                row_parts();
            } else {
                for (int j = 0; j < info->n_comp; j++) part[j] += (size_t)info->comp[j].hscale * 64;
            }
        }
        return HVC_OK;
    }

    bool finished() const { return my >= mbs_high; }

    inline void flush() {
        if (!dst) return;
        if (stream_out == 2) {
            flush_512(dst, cur);
        } else if (stream_out) {
            const __m128i zero = _mm_setzero_si128();
            for (int q = 0; q < 8; q++) {
                _mm_stream_si128((__m128i *)dst + q, _mm_load_si128((const __m128i *)cur + q));
                _mm_store_si128((__m128i *)cur + q, zero);
            }
        } else {
            std::memcpy(dst, cur, 64 * sizeof(int16_t));
            std::memset(cur, 0, 64 * sizeof(int16_t));
        }
        dst = nullptr;
    }
    void end_walk() { // the last block leaves; the streaming stores are ordered before whatever publishes the record
        flush();
        _mm_sfence();
    }
};

// Up to four look-ups (one or two AC symbols each) of the block in progress, on a LOCAL copy of the walk's hot state (bit
// reader, place in the block, table pointers: they must live in registers -- through the Walk object every symbol paid
// loads and stores of them).  The place in the block is a POINTER, BP = the block + the index of the next coefficient, END
// = the block + 64: a symbol's coefficient goes to BP[step - 1], the steps are added to BP, and BP >= END is the one
// test that ends a block.  One refill for the group; a symbol the one-lookup table does not cover is decoded by the
// two-step path after a refill of its own and ends the group.  BP becomes END at the end of the block; ERR receives the
// model's error.
#define HVC_AC_ONE_(BR, BP, END, ACW, ACF, ACT, AMAX, ERR)                                                           \
    {                                                                                                                \
        const uint64_t e_ = (ACW)[(BR).buf >> (64 - Lut::FAST_BITS)];                                                \
        const unsigned b_ = (unsigned)(e_ & 0xffu);                                                                  \
        if (!b_) {                                                                                                   \
            (BR).refill();                                                                                           \
            unsigned s_ = (ACF)[(BR).buf >> (64 - Lut::FAST_BITS)];                                                  \
            if (!s_) s_ = (ACT)[(AMAX) ? (BR).buf >> (64 - (AMAX)) : 0];                                             \
            if (!s_) {                                                                                               \
                (ERR) = HVC_E_BAD_JPEG; /* "Can't find ac code" */                                                  \
            } else {                                                                                                 \
                (BR).buf <<= s_ >> 8;                                                                                \
                (BR).cnt -= (int)(s_ >> 8);                                                                          \
                const int run_ = (s_ >> 4) & 15, size_ = s_ & 15;                                                    \
                int mag_ = 0;                                                                                        \
                if (size_) {                                                                                         \
                    mag_ = extend(size_, (unsigned)((BR).buf >> (64 - size_)));                                      \
                    (BR).buf <<= size_;                                                                              \
                    (BR).cnt -= size_;                                                                               \
                }                                                                                                    \
                if (mag_ == 0 && run_ == 0) { /* decoder.ml:131-132 (EOB, or a zero-size code) */                   \
                    (BP) = (END);                                                                                    \
                } else {                                                                                             \
                    (BP) += run_;                                                                                    \
                    if ((BP) >= (END)) (ERR) = HVC_E_BAD_JPEG; /* "coefficient index out of range" */               \
                    else *(BP)++ = (int16_t)mag_;                                                                    \
                }                                                                                                    \
            }                                                                                                        \
            break;                                                                                                   \
        }                                                                                                            \
        const uint64_t buf0_ = (BR).buf;                                                                             \
        (BR).buf <<= b_;                                                                                             \
        (BR).cnt -= (int)b_;                                                                                         \
        int16_t *const p1_ = (BP) + ((e_ >> 8) & 0xffu);  /* behind the first symbol */                              \
        int16_t *const p2_ = (BP) + ((e_ >> 16) & 0xffu); /* behind the second one (= p1_ when there is none) */     \
        p1_[-1] = (int16_t)(e_ >> 32);                    /* (past index 63: into the room behind the block) */      \
        p2_[-1] = (int16_t)(e_ >> 48);                                                                               \
        (BP) = p2_;                                                                                                  \
        if (p2_ >= (END)) { /* a last coefficient at index 63, an end of block, or an index out of range */         \
            if (p1_ >= (END)) {                                                                                      \
                if (p1_ == (END)) { /* the first symbol completed the block: what followed it is the next block's */ \
                    const unsigned b1_ = (unsigned)((e_ >> 24) & 0xffu);                                             \
                    (BR).cnt += (int)(b_ - b1_);                                                                     \
                    (BR).buf = buf0_ << b1_;                                                                         \
                } else if (p1_ - (END) < (long)Lut::WHOLE_EOB - 64) {                                                \
                    (ERR) = HVC_E_BAD_JPEG; /* "coefficient index out of range" */                                  \
                }                                                                                                    \
            } else if (p2_ > (END) && p2_ - (END) < (long)Lut::WHOLE_EOB - 64) {                                     \
                (ERR) = HVC_E_BAD_JPEG; /* "coefficient index out of range" */                                      \
            }                                                                                                        \
            (BP) = (END);                                                                                            \
            break;                                                                                                   \
        }                                                                                                            \
    }
#define HVC_AC_GROUP(BR, BP, END, ACW, ACF, ACT, AMAX, ERR)                                                          \
    do {                                                                                                             \
        (BR).refill();                                                                                               \
        HVC_AC_ONE_(BR, BP, END, ACW, ACF, ACT, AMAX, ERR)                                                           \
        HVC_AC_ONE_(BR, BP, END, ACW, ACF, ACT, AMAX, ERR)                                                           \
        HVC_AC_ONE_(BR, BP, END, ACW, ACF, ACT, AMAX, ERR)                                                           \
        HVC_AC_ONE_(BR, BP, END, ACW, ACF, ACT, AMAX, ERR)                                                           \
    } while (0)

// A segment of at most 32 bits (a file cut, or a stray marker, right behind the scan header).  Bitstream_reader.show raises
// "out of bounds" when it is asked for as many bits as the whole segment has, or more (bitstream_reader.ml:31-33) -- a
// test that says nothing about the position and that no real file meets; the windowed reader above does not make it.
// Such a segment is decoded by the model's own steps instead (decoder.ml:89-140, one `show` per code over the table of
// max_bits, one `get` per magnitude), each request held against the segment's length.
static int walk_literal(Walk &w) {
    const size_t length_in_bits = w.br.n * 8;
    size_t pos = 0;
    auto show = [&](int n, unsigned &v) -> bool { // false: the model raises
        if ((size_t)n >= length_in_bits) return false; // (so n < 32 from here on: the segment has at most 32 bits)
        v = 0;
        for (int i = 0; i < n; i++) {
            const size_t p = pos + (size_t)i;
            const unsigned bit = (p >> 3) < w.br.n ? (w.br.p[p >> 3] >> (7 - (p & 7))) & 1u : 0u; // past the end: zeros
            v = (v << 1) | bit;
        }
        return true;
    };
    while (!w.finished()) {
        // the block's place: begin_block's first half
        const Walk::McuBlock mb = w.mcu[(size_t)w.bi];
        const int i = mb.comp;
        if (!w.regular) {
            const hvc_jpeg_component &c = w.info->comp[i];
            const hvc_component &L = w.info->layout[i];
            if (w.mx * c.hscale + mb.dx >= L.blocks_w || w.my * c.vscale + mb.dy >= L.blocks_h) return HVC_E_BAD_JPEG;
        }
        int16_t *const blk = w.coefs + w.part[i] + mb.off;
        std::memset(blk, 0, 64 * sizeof(int16_t));
        unsigned code = 0, bitsv = 0;
        if (!show(w.dc[i]->max_bits, code)) return HVC_E_BAD_JPEG;
        unsigned e = w.dc[i]->e[code];
        if (!e) return HVC_E_BAD_JPEG; // "Can't find dc code"
        pos += e >> 8;
        const int cat = (int)(e & 0xff);
        long long diff = 0;
        if (cat) {
            if (!show(cat, bitsv)) return HVC_E_BAD_JPEG; // (every category of 32 bits and more ends here: Bits.get raises)
            pos += (size_t)cat;
            diff = extend_wide(cat, bitsv);
        }
        const long long dcv = add63(diff, w.dc_pred[i]);
        w.dc_pred[i] = dcv;
        if (dcv < -32768 || dcv > 32767) {
            if (!w.wide) return HVC_E_RANGE;
            try {
                w.wide->push_back(hvc::WideDc{(uint32_t)((size_t)(blk - w.coefs) >> 6), dcv});
            } catch (const std::bad_alloc &) {
                return HVC_E_OUT_OF_MEMORY;
            }
            blk[0] = (int16_t)(dcv < 0 ? -32767 : 32767);
        } else {
            blk[0] = (int16_t)dcv;
        }
        for (int k = 1; k < 64;) {
            if (!show(w.ac[i]->max_bits, code)) return HVC_E_BAD_JPEG;
            e = w.ac[i]->e[code];
            if (!e) return HVC_E_BAD_JPEG; // "Can't find ac code"
            pos += e >> 8;
            const int run = (int)((e >> 4) & 15), size = (int)(e & 15);
            int mag = 0;
            if (size) {
                if (!show(size, bitsv)) return HVC_E_BAD_JPEG;
                pos += (size_t)size;
                mag = extend(size, bitsv);
            }
            if (mag == 0 && run == 0) break; // decoder.ml:131-132
            k += run;
            if (k >= 64) return HVC_E_BAD_JPEG; // "coefficient index out of range"
            blk[k++] = (int16_t)mag;
        }
        // ... and its second half: on to the next block
        if (++w.bi == (int)w.mcu.size()) {
            w.bi = 0;
            if (++w.mx == w.mbs_wide) {
                w.mx = 0;
                ++w.my;
                w.row_parts();
            } else {
                for (int j = 0; j < w.info->n_comp; j++) w.part[j] += (size_t)w.info->comp[j].hscale * 64;
            }
        }
    }
    w.done = true;
    return HVC_OK;
}
static bool needs_literal_walk(const Walk &w) { return !w.rst_interval && w.br.n * 8 <= 32; } // (the model's own test: not with restart intervals)

// the whole file, block after block
static int walk_alone(Walk &w) {
    while (!w.finished()) {
        int r = w.begin_block();
        if (r) return r;
        BitReader br = w.br;
        int err = 0;
        int16_t *bp = w.cur + w.k, *const end = w.cur + 64;
        const uint64_t *const acw = w.acw;
        const uint16_t *const acf = w.acf, *const act = w.act;
        const int amax = w.amax;
        while (bp < end && !err) HVC_AC_GROUP(br, bp, end, acw, acf, act, amax, err);
        w.br = br;
        w.k = 64;
        if (err) return err;
    }
    w.end_walk();
    w.done = true;
    return HVC_OK;
}
} // namespace

// A walk carries its tables -- 112 KB, not something for a caller's stack (the OCaml host's threads may have small
// ones) -- and allocating it per file cost the batch pipeline's sixteen workers 2-3 % (and now and then far more, when the
// allocator trimmed and re-grew its arenas): each thread keeps two, made on first use, gone with the thread.
static Walk &thread_walk(int which) {
    struct Two { Walk w[2]; }; // (one block, like the two locals they replace)
    static thread_local std::unique_ptr<Two> t;
    if (!t) t.reset(new Two);
    return t->w[which];
}

static int entropy_decode_impl(const uint8_t *data, size_t n, const hvc_jpeg_info *info, int16_t *coefs,
                               std::vector<hvc::WideDc> *wide) {
    Walk &w = thread_walk(0);
    const int r = w.prepare(data, n, info, coefs, wide);
    return r ? r : w.done ? HVC_OK : needs_literal_walk(w) ? walk_literal(w) : walk_alone(w);
}

int hvc_jpeg_entropy_decode(const uint8_t *data, size_t n, const hvc_jpeg_info *info, int16_t *coefs) try {
    return entropy_decode_impl(data, n, info, coefs, nullptr);
} HVC_ABI_CATCH

// The same with restart intervals honoured (an extension: not the model's behaviour, see hvc::tl_honour_restart)
int hvc_jpeg_entropy_decode_restart(const uint8_t *data, size_t n, const hvc_jpeg_info *info, int16_t *coefs) try {
    hvc::RestartScope honour(true);
    return entropy_decode_impl(data, n, info, coefs, nullptr);
} HVC_ABI_CATCH

// Two files on one thread, symbol by symbol in turn; st[0] / st[1] receive each file's own status (what
// entropy_decode_impl would have returned for it): an error in one does not stop the other.
static void entropy_decode_two_impl(const uint8_t *const data[2], const size_t n[2], const hvc_jpeg_info *const info[2],
                                    int16_t *const coefs[2], std::vector<hvc::WideDc> *const wide[2], int st[2]) {
#ifdef HVC_READER_PROFILE /* experiments: where a pair of files spends its time (stderr, every 64 pairs) */
    struct Prof {
        std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now(), t1;
        ~Prof() {
            static std::atomic<long long> prep{0}, walk{0}, calls{0};
            const auto t2 = std::chrono::steady_clock::now();
            prep += std::chrono::duration_cast<std::chrono::nanoseconds>(t1 - t0).count();
            walk += std::chrono::duration_cast<std::chrono::nanoseconds>(t2 - t1).count();
            if (++calls % 64 == 0)
                std::fprintf(stderr, "reader profile: prepare %.3f ms, walk %.3f ms per pair of files\n", (double)prep / calls * 1e-6,
                             (double)walk / calls * 1e-6);
        }
    } prof;
#endif
    Walk &a = thread_walk(0), &b = thread_walk(1);
    st[0] = a.prepare(data[0], n[0], info[0], coefs[0], wide[0]);
    st[1] = b.prepare(data[1], n[1], info[1], coefs[1], wide[1]);
#ifdef HVC_READER_PROFILE
    prof.t1 = std::chrono::steady_clock::now();
#endif
    if (st[0] || st[1] || a.done || b.done || needs_literal_walk(a) || needs_literal_walk(b)) {
        // one of them cannot start, has nothing to do or is too short for the windowed reader: each runs alone
        if (!st[0] && !a.done) st[0] = needs_literal_walk(a) ? walk_literal(a) : walk_alone(a);
        if (!st[1] && !b.done) st[1] = needs_literal_walk(b) ? walk_literal(b) : walk_alone(b);
        return;
    }
    // both have a block in progress inside the hot loop; whoever ends one starts its next block and comes back
    if ((st[0] = a.begin_block()) || (st[1] = b.begin_block())) { // (the other one alone, from its first block)
        // restart whichever is intact from the top: nothing of it has been consumed beyond its first block's DC
        const int q = !st[0] ? 0 : !st[1] ? 1 : -1; // (a and b are done with: entropy_decode_impl takes the first one over)
        if (q >= 0) {
            if (wide[q]) wide[q]->clear();
            st[q] = entropy_decode_impl(data[q], n[q], info[q], coefs[q], wide[q]);
        }
        return;
    }
    BitReader bra = a.br, brb = b.br;
    int ea = 0, eb = 0;
    int16_t *bpa = a.cur + a.k, *bpb = b.cur + b.k, *const enda = a.cur + 64, *const endb = b.cur + 64;
    const uint64_t *acwa = a.acw, *acwb = b.acw;
    const uint16_t *acfa = a.acf, *acta = a.act, *acfb = b.acf, *actb = b.act;
    int amaxa = a.amax, amaxb = b.amax;
    bool alive_a = true, alive_b = true;
    while (alive_a && alive_b) {
        while (bpa < enda && bpb < endb && !(ea | eb)) {
            HVC_AC_GROUP(bra, bpa, enda, acwa, acfa, acta, amaxa, ea);
            HVC_AC_GROUP(brb, bpb, endb, acwb, acfb, actb, amaxb, eb);
        }
        if (ea || bpa >= enda) { // A: error, or its block is complete
            a.br = bra;
            a.k = 64;
            if (ea) st[0] = ea;
            if (!ea && a.finished()) a.end_walk();
            if (ea || a.finished() || (st[0] = a.begin_block())) alive_a = false;
            else {
                bra = a.br; bpa = a.cur + a.k; acwa = a.acw; acfa = a.acf; acta = a.act; amaxa = a.amax;
            }
        }
        if (eb || bpb >= endb) {
            b.br = brb;
            b.k = 64;
            if (eb) st[1] = eb;
            if (!eb && b.finished()) b.end_walk();
            if (eb || b.finished() || (st[1] = b.begin_block())) alive_b = false;
            else {
                brb = b.br; bpb = b.cur + b.k; acwb = b.acw; acfb = b.acf; actb = b.act; amaxb = b.amax;
            }
        }
    }
    // the survivor finishes alone: the block it has in progress first
    auto finish = [](Walk &w, BitReader br, int16_t *bp, int &status) {
        int err = 0;
        int16_t *const end = w.cur + 64;
        while (bp < end && !err) HVC_AC_GROUP(br, bp, end, w.acw, w.acf, w.act, w.amax, err);
        w.br = br;
        w.k = 64;
        status = err ? err : walk_alone(w);
    };
    if (alive_a) finish(a, bra, bpa, st[0]);
    if (alive_b) finish(b, brb, bpb, st[1]);
}

int hvc_jpeg_entropy_decode2(const uint8_t *jpeg_a, size_t n_a, const hvc_jpeg_info *info_a, int16_t *coefs_a, int *status_a,
                             const uint8_t *jpeg_b, size_t n_b, const hvc_jpeg_info *info_b, int16_t *coefs_b, int *status_b) try {
    if (!status_a || !status_b) return HVC_E_INVALID_ARG;
    const uint8_t *const data[2] = {jpeg_a, jpeg_b};
    const size_t n[2] = {n_a, n_b};
    const hvc_jpeg_info *const info[2] = {info_a, info_b};
    int16_t *const coefs[2] = {coefs_a, coefs_b};
    std::vector<hvc::WideDc> *const wide[2] = {nullptr, nullptr};
    int st[2] = {HVC_OK, HVC_OK};
    entropy_decode_two_impl(data, n, info, coefs, wide, st);
    *status_a = st[0];
    *status_b = st[1];
    return HVC_OK;
} HVC_ABI_CATCH

// Decoder.crop (decoder.ml:403-413) of the first `n_planes` components: the actual_w x actual_h top-left part of each
// padded plane, planes back to back.  (the caller's info: every crop must lie inside the plane it is cut from)
static bool crops_are_inside(const hvc_jpeg_info *info) {
    for (int i = 0; i < info->n_comp; i++) {
        const hvc_jpeg_component &c = info->comp[i];
        const hvc_component &L = info->layout[i];
        if (c.actual_width < 0 || c.actual_height < 0 || c.actual_width > c.decoded_width || c.actual_height > c.decoded_height ||
            L.stride < (size_t)c.decoded_width || L.plane_offset > info->pixel_bytes ||
            (c.decoded_height > 0 && c.decoded_width > 0 &&
             ((size_t)c.decoded_height - 1) * L.stride + (size_t)c.decoded_width > info->pixel_bytes - L.plane_offset))
            return false;
    }
    return true;
}
static int crop_planes(const hvc_jpeg_info *info, int n_planes, const uint8_t *pixels, uint8_t *out, size_t cap, size_t *out_len) {
    size_t need = 0;
    for (int i = 0; i < n_planes; i++) need += (size_t)info->comp[i].actual_width * (size_t)info->comp[i].actual_height;
    if (out_len) *out_len = need;
    if (need > cap) return HVC_E_INVALID_ARG;
    if (need && (!pixels || !out)) return HVC_E_INVALID_ARG; // (planes without a sample need no memory)
    uint8_t *o = out;
    for (int i = 0; i < n_planes; i++) {
        const hvc_jpeg_component &c = info->comp[i];
        if (c.actual_width == 0) continue;
        const uint8_t *p = pixels + info->layout[i].plane_offset;
        for (int y = 0; y < c.actual_height; y++, o += c.actual_width)
            std::memcpy(o, p + (size_t)y * info->layout[i].stride, (size_t)c.actual_width);
    }
    return HVC_OK;
}

// Decoder.get_yuv_frame (decoder.ml:415-420) = Frame.of_planes of the crops of components 0, 1 and 2 -- and of_planes
// (common/src/frame.ml:42-61) makes a Frame.t only of planes it can name: `components.(1)` / `.(2)` must exist (Invalid_argument
// otherwise), the chroma planes must be of one size ("Chroma planes must be same width and height") and that size must be
// the luma plane's halved both ways, halved in width, or the same (C420 / C422 / C444, tried in this order, integer
// halves; "Could not infer chroma subsampling").  Everything else the model has decoded -- 4:1:1, 4:4:0, one or two
// components, a plane without samples beside planes with -- it cannot hand out as a frame: HVC_E_BAD_JPEG, the model's raise.
// A fourth component is decoded and left out, as there.  Output: the three crops back to back (Frame.output order, frame.ml:66-70).
int hvc_jpeg_get_yuv_frame(const hvc_jpeg_info *info, const uint8_t *pixels, uint8_t *out, size_t cap, size_t *out_len) try {
    if (!info || info->n_comp < 1 || info->n_comp > 4 || !crops_are_inside(info)) return HVC_E_INVALID_ARG;
    if (out_len) *out_len = 0;
    if (info->n_comp < 3) return HVC_E_BAD_JPEG;
    const hvc_jpeg_component &y = info->comp[0], &u = info->comp[1], &v = info->comp[2];
    if (u.actual_width != v.actual_width || u.actual_height != v.actual_height) return HVC_E_BAD_JPEG;
    const bool c420 = y.actual_width / 2 == u.actual_width && y.actual_height / 2 == u.actual_height;
    const bool c422 = y.actual_width / 2 == u.actual_width && y.actual_height == u.actual_height;
    const bool c444 = y.actual_width == u.actual_width && y.actual_height == u.actual_height;
    if (!c420 && !c422 && !c444) return HVC_E_BAD_JPEG;
    return crop_planes(info, 3, pixels, out, cap, out_len);
} HVC_ABI_CATCH

// Array.map crop over Decoder.get_decoded_planes (decoder.ml:399-413): EVERY component's crop, back to back in scan order,
// whatever the sampling -- what a caller takes where Frame.of_planes has no name for the planes (4:1:1, 4:4:0, grey, four
// components).  For a frame get_yuv_frame accepts, the same bytes (plus the fourth component's, if there is one).
int hvc_jpeg_get_cropped_planes(const hvc_jpeg_info *info, const uint8_t *pixels, uint8_t *out, size_t cap, size_t *out_len) try {
    if (!info || info->n_comp < 1 || info->n_comp > 4 || !crops_are_inside(info)) return HVC_E_INVALID_ARG;
    return crop_planes(info, info->n_comp, pixels, out, cap, out_len);
} HVC_ABI_CATCH

} // extern "C"

// ===========================================================================
// Encoder back end
namespace {

// ITU-T T.81 Annex K.3 tables = Tables.Default (tables.ml:54-476), BITS / HUFFVAL form
const uint8_t K_DC_LUMA_BITS[16] = {0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
const uint8_t K_DC_CHROMA_BITS[16] = {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
const uint8_t K_DC_VALS[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
const uint8_t K_AC_LUMA_BITS[16] = {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d};
const uint8_t K_AC_CHROMA_BITS[16] = {0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77};
const uint8_t K_AC_LUMA_VALS[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71,
    0x14, 0x32, 0x81, 0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72,
    0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37,
    0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59,
    0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83,
    0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3,
    0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3,
    0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2,
    0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
const uint8_t K_AC_CHROMA_VALS[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22,
    0x32, 0x81, 0x08, 0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1,
    0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36,
    0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58,
    0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a,
    0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a,
    0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba,
    0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda,
    0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};

// jpeg/model/src/quant_tables.ml:3-137 (Annex K numbers in array order, used as zig-zag order)
const uint8_t K_Q_LUMA[64] = {16, 11, 10, 16, 24,  40,  51,  61,  12, 12, 14, 19, 26,  58,  60,  55,
                              14, 13, 16, 24, 40,  57,  69,  56,  14, 17, 22, 29, 51,  87,  80,  62,
                              18, 22, 37, 56, 68,  109, 103, 77,  24, 35, 55, 64, 81,  104, 113, 92,
                              49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
const uint8_t K_Q_CHROMA[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99,
                                99, 99, 47, 66, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
                                99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};

struct EncCode { uint16_t bits; uint8_t len; };
// ... and as the coder wants them: the code already shifted past the magnitude bits, the magnitude's mask, code + magnitude
// bits -- a field is `hi | (magnitude & mask)`, `total` bits long
struct EncField { uint32_t hi; uint16_t mask; uint8_t total, pad; };
struct EncTable { EncCode dc[16]; EncCode ac[256]; EncField dcf[16]; EncField acf[256]; }; // ac indexed by (run << 4) | size

void build_enc(EncTable &t, const uint8_t *dc_bits, const uint8_t *dc_vals, const uint8_t *ac_bits, const uint8_t *ac_vals) {
    std::memset(&t, 0, sizeof t);
    unsigned code = 0;
    int k = 0;
    for (int len = 1; len <= 16; len++) { // canonical code assignment, tables.ml:27-45
        for (int i = 0; i < dc_bits[len - 1]; i++, k++) t.dc[dc_vals[k]] = EncCode{(uint16_t)(code + i), (uint8_t)len};
        code = (code + dc_bits[len - 1]) << 1;
    }
    code = 0;
    k = 0;
    for (int len = 1; len <= 16; len++) {
        for (int i = 0; i < ac_bits[len - 1]; i++, k++) t.ac[ac_vals[k]] = EncCode{(uint16_t)(code + i), (uint8_t)len};
        code = (code + ac_bits[len - 1]) << 1;
    }
    for (int s = 0; s < 16; s++) // (a DC category above 11 / an AC size above 10 has no code in the default tables: refused before use)
        t.dcf[s] = EncField{(uint32_t)t.dc[s].bits << s, (uint16_t)((1u << s) - 1u), (uint8_t)(t.dc[s].len + s), 0};
    for (int rs = 0; rs < 256; rs++) {
        const int s = rs & 15;
        t.acf[rs] = EncField{(uint32_t)t.ac[rs].bits << s, (uint16_t)((1u << s) - 1u), (uint8_t)(t.ac[rs].len + s), 0};
    }
}

// Bitstream_writer (common/src/bitstream_writer.ml): MSB first, 0xff -> 0xff00 stuffing -- in two stages.
// Stage 1, per field: the bits go to a scratch buffer WITHOUT stuffing and without a branch -- the accumulator is kept
// MSB-aligned, its eight bytes are stored at the cursor every time (whatever of them is not complete yet is stored
// again by the next field), and the cursor moves on by the bytes that are complete.  Stage 2, per MCU row: the complete
// bytes are appended to the output with a 0x00 behind every 0xff, sixteen at a time where there is none.
// (Round 2's writer tested the accumulator for a full 32-bit word and that word for 0xff bytes behind every field: a
// branch mispredicted every fifth field.)
struct BitWriter {
    uint8_t *cur;     // scratch cursor: the next byte that is not complete yet
    uint64_t acc = 0; // the bits of that byte and what follows, MSB-aligned
    int nbits = 0;    // how many of them are valid (< 8 between fields)
    explicit BitWriter(uint8_t *scratch) : cur(scratch) {}
    inline void put(unsigned value, int bits) { // 1 <= bits <= 27, value < 2^bits; needs 8 writable bytes at the cursor
        acc |= (uint64_t)value << (64 - nbits - bits);
        nbits += bits;
        const uint64_t be = __builtin_bswap64(acc);
        std::memcpy(cur, &be, 8);
        const int whole = nbits & ~7;
        cur += whole >> 3;
        acc <<= whole; // (whole <= 32)
        nbits &= 7;
    }
    void pad_with_1s() { // Bitstream_writer.flush_with_1s (bitstream_writer.ml:45-49)
        if (nbits) put((1u << (8 - nbits)) - 1u, 8 - nbits);
    }
};

// scratch[0, n) -> appended to o with the stuffing (stage 2)
__attribute__((target("sse2"))) static void append_stuffed(std::vector<uint8_t> &o, const uint8_t *src, size_t n) {
    const size_t at = o.size();
    o.resize(at + 2 * n); // (every byte an 0xff: trimmed below)
    uint8_t *d = o.data() + at;
    const __m128i ff = _mm_set1_epi8((char)0xff);
    size_t i = 0;
    while (i < n) {
        if (i + 16 <= n) {
            const __m128i v = _mm_loadu_si128(reinterpret_cast<const __m128i *>(src + i));
            if (!_mm_movemask_epi8(_mm_cmpeq_epi8(v, ff))) {
                _mm_storeu_si128(reinterpret_cast<__m128i *>(d), v);
                d += 16;
                i += 16;
                continue;
            }
        }
        const size_t stop = i + 16 <= n ? i + 16 : n;
        for (; i < stop; i++) {
            const uint8_t b = src[i];
            *d++ = b;
            if (b == 0xff) *d++ = 0;
        }
    }
    o.resize((size_t)(d - o.data()));
}

// bit k set <=> q[k] != 0
inline uint64_t nonzero_mask(const int16_t *q) {
    const __m128i z = _mm_setzero_si128();
    uint64_t m = 0;
    for (int i = 0; i < 4; i++) {
        const __m128i a = _mm_loadu_si128(reinterpret_cast<const __m128i *>(q + 16 * i));
        const __m128i b = _mm_loadu_si128(reinterpret_cast<const __m128i *>(q + 16 * i + 8));
        const unsigned zero16 = (unsigned)_mm_movemask_epi8(_mm_packs_epi16(_mm_cmpeq_epi16(a, z), _mm_cmpeq_epi16(b, z)));
        m |= (uint64_t)(~zero16 & 0xffffu) << (16 * i);
    }
    return m;
}

inline int bit_size(int v) { // encoder.ml:143
    const unsigned a = (unsigned)(v < 0 ? -v : v);
    return a ? 32 - __builtin_clz(a) : 0;
}

void put_marker(std::vector<uint8_t> &o, int code) { o.push_back(0xff); o.push_back((uint8_t)code); }
void put16(std::vector<uint8_t> &o, int v) { o.push_back((uint8_t)(v >> 8)); o.push_back((uint8_t)v); }

void write_dht(std::vector<uint8_t> &o, int tclass, int id, const uint8_t *bits, const uint8_t *vals) {
    int total = 0;
    for (int i = 0; i < 16; i++) total += bits[i];
    put_marker(o, 0xc4);
    put16(o, 3 + 16 + total);
    o.push_back((uint8_t)((tclass << 4) | id));
    o.insert(o.end(), bits, bits + 16);
    o.insert(o.end(), vals, vals + total);
}

} // namespace

extern "C" {

// Quant_tables.scale (quant_tables.ml:139-147)
int hvc_quant_table(int chroma_table, int quality, uint16_t *out) try {
    if (!out) return HVC_E_INVALID_ARG;
    int q = quality < 1 ? 1 : (quality > 100 ? 100 : quality);
    const int s = q < 50 ? 5000 / q : 200 - 2 * q;
    const uint8_t *t = chroma_table ? K_Q_CHROMA : K_Q_LUMA;
    for (int i = 0; i < 64; i++) {
        int d = (t[i] * s + 50) / 100;
        out[i] = (uint16_t)(d < 1 ? 1 : (d > 255 ? 255 : d));
    }
    return HVC_OK;
} HVC_ABI_CATCH

// Encoder.Parameters.c420/c422/c444 (encoder.ml:347-349) + Encoder.create plane geometry (:437-472):
// the padded plane layout the encode kernel reads and the coefficient record it writes.
int hvc_jpeg_encoder_layout(int width, int height, int chroma, int quality, hvc_jpeg_info *info) try {
    if (!info || width < 1 || height < 1 || width > 65535 || height > 65535) return HVC_E_INVALID_ARG;
    static const int S420[6] = {2, 2, 1, 1, 1, 1}, S422[6] = {2, 2, 1, 2, 1, 2}, S444[6] = {1, 1, 1, 1, 1, 1};
    const int *s = chroma == 420 ? S420 : chroma == 422 ? S422 : chroma == 444 ? S444 : nullptr;
    if (!s) return HVC_E_INVALID_ARG;
    std::memset(info, 0, sizeof *info);
    info->width = width;
    info->height = height;
    info->n_comp = 3;
    info->n_qtabs = 2;
    hvc_quant_table(0, quality, info->qtabs[0]);
    hvc_quant_table(1, quality, info->qtabs[1]);
    const int max_h = s[0], max_v = s[1]; // luma carries the maxima in all three parameter sets
    size_t coef_off = 0, pix_off = 0;
    for (int i = 0; i < 3; i++) {
        hvc_jpeg_component &c = info->comp[i];
        c.identifier = i + 1;
        c.hscale = s[2 * i];
        c.vscale = s[2 * i + 1];
        c.actual_width = width * c.hscale / max_h;
        c.actual_height = height * c.vscale / max_v;
        c.decoded_width = (c.actual_width + 8 * c.hscale - 1) / (8 * c.hscale) * (8 * c.hscale);
        c.decoded_height = (c.actual_height + 8 * c.vscale - 1) / (8 * c.vscale) * (8 * c.vscale);
        c.dc_table = c.ac_table = i ? 1 : 0;
        hvc_component &L = info->layout[i];
        L.blocks_w = c.decoded_width / 8;
        L.blocks_h = c.decoded_height / 8;
        L.qtab = i ? 1 : 0;
        L.coef_offset = coef_off;
        L.plane_offset = pix_off;
        L.stride = (size_t)c.decoded_width;
        coef_off += (size_t)L.blocks_w * L.blocks_h * 64;
        pix_off += (size_t)c.decoded_width * c.decoded_height;
    }
    info->coef_count = coef_off;
    info->pixel_bytes = pix_off;
    return HVC_OK;
} HVC_ABI_CATCH

// Encoder.write_headers (encoder.ml:371-418) + encode_seq's rle / write_bits (:127-193, 476-505) +
// complete_and_write_eoi (:507-510) over one frame's quantised coefficient record (zig-zag, DC absolute).
// encode_seq (encoder.ml:476-505) walks the MCU grid of component 0 and reads h x v blocks of every
// component per MCU; where that grid reaches past a component's plane the model raises
// "[Plane.get] out of bounds" (plane.ml:43-50): 4:2:0 / 4:2:2 at width 16k + 1 (or height 16k + 1).
int hvc_jpeg_encoder_check(const hvc_jpeg_info *info) try {
    if (!info || info->n_comp != 3 || !info_is_sane(info)) return HVC_E_INVALID_ARG;
    const hvc_jpeg_component &c0 = info->comp[0];
    const int mbs_wide = c0.decoded_width / (8 * c0.hscale), mbs_high = c0.decoded_height / (8 * c0.vscale);
    for (int i = 0; i < 3; i++)
        if (mbs_wide * info->comp[i].hscale > info->layout[i].blocks_w ||
            mbs_high * info->comp[i].vscale > info->layout[i].blocks_h)
            return HVC_E_INVALID_ARG;
    // The encoder writes the two default table sets only (write_dht, encoder.ml:236-264: luma = 0, chroma = 1) and
    // codes a component's DC and AC symbols with the same set (Component.t, encoder.ml:287-345); the header holds two
    // quantiser tables.  An info from hvc_jpeg_read_header may carry other selectors: refused, not guessed at.
    for (int i = 0; i < 3; i++) {
        const int dt = info->comp[i].dc_table, at = info->comp[i].ac_table, qt = info->layout[i].qtab;
        if (dt < 0 || dt > 1 || at != dt || qt < 0 || qt > 1) return HVC_E_INVALID_ARG;
    }
    return HVC_OK;
} HVC_ABI_CATCH

} // extern "C"

namespace hvc {
int entropy_decode_wide(const uint8_t *data, size_t n, const ::hvc_jpeg_info *info, int16_t *coefs, std::vector<WideDc> &wide) {
    wide.clear();
    return entropy_decode_impl(data, n, info, coefs, &wide);
}
// two files on one thread (entropy_decode_two_impl): the batch pipeline's workers
void entropy_decode_wide2(const uint8_t *const data[2], const size_t n[2], const ::hvc_jpeg_info *const info[2], int16_t *const coefs[2],
                          std::vector<WideDc> *const wide[2], int st[2]) {
    wide[0]->clear();
    wide[1]->clear();
    entropy_decode_two_impl(data, n, info, coefs, wide, st);
}

// Encoder.write_headers (encoder.ml:371-418) for `info`: SOI .. SOS, appended to o
void jpeg_header_bytes(const hvc_jpeg_info *info, std::vector<uint8_t> &o) {
    put_marker(o, 0xd8);
    { // write_app0 "Hardcaml JPEG."
        static const char tag[] = "Hardcaml JPEG.";
        put_marker(o, 0xe0);
        put16(o, 2 + (int)sizeof(tag) - 1);
        o.insert(o.end(), tag, tag + sizeof(tag) - 1);
    }
    for (int t = 0; t < 2; t++) { // write_dqt, element_precision 8
        put_marker(o, 0xdb);
        put16(o, 3 + 64);
        o.push_back((uint8_t)t);
        for (int i = 0; i < 64; i++) o.push_back((uint8_t)info->qtabs[t][i]);
    }
    put_marker(o, 0xc0); // write_sof
    put16(o, 2 + 6 + 3 * 3);
    o.push_back(8);
    put16(o, info->height);
    put16(o, info->width);
    o.push_back(3);
    for (int i = 0; i < 3; i++) {
        o.push_back((uint8_t)info->comp[i].identifier);
        o.push_back((uint8_t)((info->comp[i].hscale << 4) | info->comp[i].vscale));
        o.push_back((uint8_t)info->layout[i].qtab);
    }
    write_dht(o, 0, 0, K_DC_LUMA_BITS, K_DC_VALS);
    write_dht(o, 0, 1, K_DC_CHROMA_BITS, K_DC_VALS);
    write_dht(o, 1, 0, K_AC_LUMA_BITS, K_AC_LUMA_VALS);
    write_dht(o, 1, 1, K_AC_CHROMA_BITS, K_AC_CHROMA_VALS);
    put_marker(o, 0xda); // write_sos
    put16(o, 2 + 4 + 3 * 2);
    o.push_back(3);
    for (int i = 0; i < 3; i++) {
        o.push_back((uint8_t)info->comp[i].identifier);
        o.push_back((uint8_t)((info->comp[i].dc_table << 4) | info->comp[i].ac_table));
    }
    o.push_back(0);
    o.push_back(63);
    o.push_back(0);

}

// the default Huffman tables as the GPU coder wants them: per table set t (0 luma, 1 chroma)
// out[t][size] for DC (12 entries) and out[t][16 + ((run << 4) | size)] for AC, each (code << 5) | length
void default_enc_tables(uint32_t (*out)[16 + 256]) {
    EncTable et[2];
    build_enc(et[0], K_DC_LUMA_BITS, K_DC_VALS, K_AC_LUMA_BITS, K_AC_LUMA_VALS);
    build_enc(et[1], K_DC_CHROMA_BITS, K_DC_VALS, K_AC_CHROMA_BITS, K_AC_CHROMA_VALS);
    for (int t = 0; t < 2; t++) {
        for (int i = 0; i < 16; i++) out[t][i] = ((uint32_t)et[t].dc[i].bits << 5) | et[t].dc[i].len;
        for (int i = 0; i < 256; i++) out[t][16 + i] = ((uint32_t)et[t].ac[i].bits << 5) | et[t].ac[i].len;
    }
}

// ECS extraction shared by the host decoder and the GPU decoder: extract_entropy_coded_bits
// (decoder.ml:261-281): up to the first marker, 0xff00 -> 0xff
// The entropy-coded segment without its stuffing (0xFF 0x00 -> 0xFF), up to the first marker, into dst[0, cap):
// the number of bytes, or SIZE_MAX when cap is too small.
// 16 bytes at a time (SSE2 compares, SSSE3 byte shuffle): the stuffed zeros -- a 0x00 right behind an 0xFF -- are
// squeezed out through a table of shuffles, 8 bytes per look-up.  A chunk holding a marker (0xFF followed by anything
// but 0x00) or too close to the end of the data is left to the byte-wise loop behind it.  A low-quality 1080p file has an
// 0xFF every ~20 bytes (10 795 of them in 228 kB at quality 3): the memchr + memcpy pair per run that stood here cost a
// single file's call 90 of its 650 us, and the batch pipelines' workers the same per file.
struct UnstuffLut {
    alignas(16) uint8_t shuf[256][8]; // [mask of bytes to keep]: their positions, packed to the front (0x80 = zero fill)
    UnstuffLut() {
        for (int m = 0; m < 256; m++) {
            int k = 0;
            for (int i = 0; i < 8; i++)
                if (m & (1 << i)) shuf[m][k++] = (uint8_t)i;
            while (k < 8) shuf[m][k++] = 0x80;
        }
    }
};
__attribute__((target("ssse3,popcnt"))) static void unstuff_chunks(const uint8_t *data, size_t n, size_t &pos, uint8_t *dst, size_t cap,
                                                                   size_t &out) {
    static const UnstuffLut lut;
    const __m128i ff16 = _mm_set1_epi8((char)0xff), zero16 = _mm_setzero_si128();
    unsigned prev_ff = 0; // the byte in front of the chunk was an 0xFF that belongs to the data
    while (pos + 17 <= n && out + 16 <= cap) { // (+ 1: the byte behind the chunk decides about an 0xFF in its last place)
        const __m128i v = _mm_loadu_si128(reinterpret_cast<const __m128i *>(data + pos));
        const unsigned m_ff = (unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(v, ff16));
        if (!(m_ff | prev_ff)) {
            _mm_storeu_si128(reinterpret_cast<__m128i *>(dst + out), v);
            pos += 16;
            out += 16;
            continue;
        }
        const unsigned m_zero = (unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(v, zero16));
        const unsigned next_zero = data[pos + 16] == 0x00 ? 1u : 0u;
        // an 0xFF whose successor is not 0x00 is a marker (decoder.ml:261-281 stops there): the byte-wise loop takes over
        if (m_ff & ~((m_zero >> 1) | (next_zero << 15)) & 0xffffu) break;
        const unsigned drop = m_zero & ((m_ff << 1) | prev_ff) & 0xffffu; // zeros right behind an 0xFF
        const unsigned keep = ~drop & 0xffffu;
        const __m128i lo = _mm_shuffle_epi8(v, _mm_loadl_epi64(reinterpret_cast<const __m128i *>(lut.shuf[keep & 0xffu])));
        const __m128i hi = _mm_shuffle_epi8(_mm_srli_si128(v, 8), _mm_loadl_epi64(reinterpret_cast<const __m128i *>(lut.shuf[keep >> 8])));
        _mm_storel_epi64(reinterpret_cast<__m128i *>(dst + out), lo);
        out += (size_t)__builtin_popcount(keep & 0xffu);
        _mm_storel_epi64(reinterpret_cast<__m128i *>(dst + out), hi); // (8 bytes each: out + 16 <= cap was checked)
        out += (size_t)__builtin_popcount(keep >> 8);
        pos += 16;
        prev_ff = (m_ff >> 15) & 1u;
    }
    // hand over on a boundary the byte-wise loop understands: it expects to look at an 0xFF itself, so if the chunk
    // loop stopped right behind one (prev_ff), give that byte back
    if (prev_ff) {
        pos -= 1;
        out -= 1;
    }
}

size_t extract_ecs_to(const uint8_t *data, size_t n, size_t pos, uint8_t *dst, size_t cap) {
    size_t out = 0;
    static const bool simd = __builtin_cpu_supports("ssse3") && __builtin_cpu_supports("popcnt");
    while (pos < n) {
        if (simd) unstuff_chunks(data, n, pos, dst, cap, out);
        if (pos >= n) break;
        // byte-wise from here to the next 0xFF pair (the end of the segment, a marker, or data the chunk loop left)
        const uint8_t *ff = (const uint8_t *)std::memchr(data + pos, 0xff, n - pos);
        const size_t stop = ff ? (size_t)(ff - data) : n;
        if (stop - pos > cap - out) return SIZE_MAX;
        std::memcpy(dst + out, data + pos, stop - pos);
        out += stop - pos;
        if (!ff) break;
        const int next = stop + 1 < n ? data[stop + 1] : 0;
        if (next != 0x00) break;
        if (out == cap) return SIZE_MAX;
        dst[out++] = 0xff;
        pos = stop + 2;
        if (n - pos < 17 || cap - out < 16) { // too little left for the chunk loop: finish byte-wise
            while (pos < n) {
                const uint8_t *f2 = (const uint8_t *)std::memchr(data + pos, 0xff, n - pos);
                const size_t st2 = f2 ? (size_t)(f2 - data) : n;
                if (st2 - pos > cap - out) return SIZE_MAX;
                std::memcpy(dst + out, data + pos, st2 - pos);
                out += st2 - pos;
                if (!f2) return out;
                const int nx = st2 + 1 < n ? data[st2 + 1] : 0;
                if (nx != 0x00) return out;
                if (out == cap) return SIZE_MAX;
                dst[out++] = 0xff;
                pos = st2 + 2;
            }
            break;
        }
    }
    return out;
}
static void extract_ecs(const uint8_t *data, size_t n, size_t pos, std::vector<uint8_t> &ecs) {
    ecs.resize((n > pos ? n - pos : 0) + 1); // unstuffing only ever shortens
    const size_t got = extract_ecs_to(data, n, pos, ecs.data(), ecs.size());
    ecs.resize(got == SIZE_MAX ? 0 : got);
}

static int prepare_gpu_decode_impl(const uint8_t *jpeg, size_t n, const ::hvc_jpeg_info *info, HdTables &t,
                                   std::vector<uint8_t> *ecs, uint8_t *dst, size_t cap, size_t *ecs_size, bool &gpu_ok, RstUnits *units = nullptr);
int prepare_gpu_decode(const uint8_t *jpeg, size_t n, const ::hvc_jpeg_info *info, HdTables &t, std::vector<uint8_t> &ecs,
                       bool &gpu_ok) {
    return prepare_gpu_decode_impl(jpeg, n, info, t, &ecs, nullptr, 0, nullptr, gpu_ok);
}
int prepare_gpu_decode_to(const uint8_t *jpeg, size_t n, const ::hvc_jpeg_info *info, HdTables &t, uint8_t *dst, size_t cap,
                          size_t *ecs_size, bool &gpu_ok, RstUnits *units) {
    return prepare_gpu_decode_impl(jpeg, n, info, t, nullptr, dst, cap, ecs_size, gpu_ok, units);
}
unsigned restart_interval_of(const uint8_t *jpeg, size_t n) {
    Header h;
    return parse_header(jpeg, n, h) ? 0u : (unsigned)h.restart_interval;
}
// Restart mode: every interval unstuffed into its own slot of dst -- interval k at off[k] (a multiple of 16), len[k] bytes,
// zeros behind it to the end of its slot, hd_unit_slot(len[k]) bytes.  Follows extract_ecs_restart marker for marker.
// Returns the number of intervals found, or SIZE_MAX when there are more than `most` or dst is too small.
static size_t extract_ecs_units_to(const uint8_t *data, size_t n, size_t pos, uint8_t *dst, size_t cap, unsigned most, unsigned *off,
                                   unsigned *len, size_t *total) {
    size_t at = 0, out = 0, k = 0; // the open interval's slot starts at `at`, `out` bytes of it are written
    *total = 0;
    auto close = [&]() -> bool {
        const size_t slot = hd_unit_slot(out);
        if (k >= most || slot > cap - at) return false; // (the bytes themselves fitted: checked as they were written)
        std::memset(dst + at + out, 0, slot - out);
        off[k] = (unsigned)at;
        len[k] = (unsigned)out;
        *total += out;
        k++;
        at += slot;
        out = 0;
        return true;
    };
    while (pos < n) {
        const uint8_t *ff = (const uint8_t *)std::memchr(data + pos, 0xff, n - pos);
        const size_t stop = ff ? (size_t)(ff - data) : n;
        if (at > cap || stop - pos > cap - at - out) return SIZE_MAX;
        std::memcpy(dst + at + out, data + pos, stop - pos);
        out += stop - pos;
        if (!ff) break;
        const int next = stop + 1 < n ? data[stop + 1] : -1;
        if (next == 0x00) {
            if (out >= cap - at) return SIZE_MAX;
            dst[at + out++] = 0xff;
            pos = stop + 2;
        } else if (next == 0xff) {
            pos = stop + 1;
        } else if (next >= 0xd0 && next <= 0xd7) {
            if (!close()) return SIZE_MAX;
            pos = stop + 2;
        } else {
            break;
        }
    }
    if (at > cap || !close()) return SIZE_MAX;
    return k;
}
static int prepare_gpu_decode_impl(const uint8_t *jpeg, size_t n, const ::hvc_jpeg_info *info, HdTables &t,
                                   std::vector<uint8_t> *ecs, uint8_t *dst, size_t cap, size_t *ecs_size, bool &gpu_ok, RstUnits *units) {
    gpu_ok = false;
    Header h;
    int r = parse_header(jpeg, n, h);
    if (r) return r;
    std::memset(&t, 0, sizeof t);
    auto fill = [&](const HuffSpec &s, HdTable &o, HdOvfRaw &ov) -> bool {
        unsigned code = 0;
        int k = 0, nsub = 0;
        unsigned long long kraft = 0; // in units of 2^-16
        for (int len = 1; len <= 16; len++) {
            if (len > 10) { // canonical form of the long codes (the overflow search; unused unless a prefix overflows)
                ov.mincode[len - 11] = (uint16_t)code;
                ov.count[len - 11] = (uint16_t)s.lengths[len - 1];
                ov.valptr[len - 11] = (uint16_t)k;
            }
            for (int i = 0; i < s.lengths[len - 1]; i++, k++) {
                if (k >= 256) return false;
                const unsigned cw = code + (unsigned)i; // canonical code of this symbol, `len` bits
                const uint16_t entry = (uint16_t)((len << 8) | s.values[k]);
                ov.vals[k] = (uint8_t)s.values[k];
                ov.lens[k] = (uint8_t)len;
                if (len <= 10) {
                    const unsigned f0 = cw << (10 - len), fc = 1u << (10 - len);
                    for (unsigned j = 0; j < fc && f0 + j < 1024u; j++) o.fast[f0 + j] = entry;
                } else {
                    const unsigned prefix = cw >> (len - 10);
                    if (prefix >= 1024u) return false;
                    if (!(o.fast[prefix] & 0x8000u)) {
                        if (o.fast[prefix] != 0) return false; // no prefix code
                        // the ninth prefix and those after it: no sub-table, the canonical search finds their symbols
                        o.fast[prefix] = (uint16_t)(0x8000u | (nsub < HVC_HD_SUBTABLES ? (unsigned)nsub++ : HVC_HD_OVF));
                        if ((o.fast[prefix] & 0x7fffu) == HVC_HD_OVF) ov.used = 1;
                    }
                    const unsigned sub = o.fast[prefix] & 0x7fffu;
                    if (sub == HVC_HD_OVF) continue;
                    const unsigned rest = cw & ((1u << (len - 10)) - 1u); // the len - 10 bits after the prefix
                    const unsigned f0 = rest << (16 - len), fc = 1u << (16 - len);
                    for (unsigned j = 0; j < fc; j++) o.sub[sub * 64 + f0 + j] = entry;
                }
            }
            kraft += (unsigned long long)s.lengths[len - 1] << (16 - len);
            code = (code + (unsigned)s.lengths[len - 1]) << 1;
        }
        return kraft <= (1ull << 16); // a prefix code (possibly incomplete): the two-level table equals the model's LUT
    };
    bool ok = info->n_comp <= 3;
    for (int i = 0; i < info->n_comp; i++) {
        int di = -1, ai = -1; // find_huffman_table (decoder.ml:238-259): newest match
        for (int k = (int)h.dht.size() - 1; k >= 0; k--) {
            if (di < 0 && h.dht[k].tclass == 0 && h.dht[k].id == info->comp[i].dc_table) di = k;
            if (ai < 0 && h.dht[k].tclass == 1 && h.dht[k].id == info->comp[i].ac_table) ai = k;
        }
        if (di < 0 || ai < 0) return HVC_E_BAD_JPEG;
        Lut probe; // the host decoder's own validity check
        if (!probe.build(h.dht[di].spec) || !probe.build(h.dht[ai].spec)) return HVC_E_BAD_JPEG;
        if (i < 3) {
            ok = ok && fill(h.dht[di].spec, t.dc[i], t.ovf_dc[i]);
            ok = ok && fill(h.dht[ai].spec, t.ac[i], t.ovf_ac[i]);
        }
    }
    // Restart intervals honoured (opt-in) and the scan has more than one: every interval is a unit of its own for the GPU
    // reader (hvc_hdec.h HdParams::rst_*) -- for a caller that lays units out; any other leaves the file to the host reader.
    // (One interval = no marker is expected: the plain segment, which ends at the first marker whatever it is.)
    unsigned long long mcus = 0;
    if (info->n_comp > 0 && info->comp[0].hscale > 0 && info->comp[0].vscale > 0)
        mcus = (unsigned long long)(info->comp[0].decoded_width / (8 * info->comp[0].hscale)) *
               (unsigned long long)(info->comp[0].decoded_height / (8 * info->comp[0].vscale));
    const bool rst_multi = tl_honour_restart && h.restart_interval > 0 && mcus > (unsigned long long)h.restart_interval;
    size_t got;
    if (rst_multi) {
        const unsigned long long ipf = (mcus + (unsigned)h.restart_interval - 1) / (unsigned)h.restart_interval;
        got = 0;
        if (ecs_size) *ecs_size = 0;
        if (!units || ecs || !dst || units->ipf != ipf || units->interval != (unsigned)h.restart_interval) {
            ok = false;
        } else {
            const size_t found = extract_ecs_units_to(jpeg, n, h.ecs_pos, dst, cap, units->ipf, units->off, units->len, &got);
            ok = ok && found == units->ipf; // (fewer or more markers than the DRI promises: the host reader's)
            if (ecs_size) *ecs_size = got;
        }
    } else if (units) { // (a batch laid out for intervals, a file without them)
        ok = false;
        got = 0;
        if (ecs_size) *ecs_size = 0;
    } else if (ecs) {
        extract_ecs(jpeg, n, h.ecs_pos, *ecs);
        got = ecs->size();
    } else {
        got = extract_ecs_to(jpeg, n, h.ecs_pos, dst, cap); // SIZE_MAX: does not fit (the caller sized cap from the file sizes)
        *ecs_size = got == SIZE_MAX ? 0 : got;
        ok = ok && got != SIZE_MAX;
    }
    int per_mcu = 0;
    bool empty_plane = info->coef_count == 0; // (the model's empty planes: the host reader walks around them)
    for (int i = 0; i < info->n_comp; i++) {
        per_mcu += info->comp[i].hscale * info->comp[i].vscale;
        empty_plane |= info->comp[i].hscale < 1 || info->comp[i].vscale < 1;
    }
    // (a segment of at most 32 bits: the host reader has the model's length test for those -- walk_literal)
    gpu_ok = ok && !empty_plane && info->n_comp <= 3 && per_mcu <= HVC_HD_MAX_MCU_BLOCKS && got < (1u << 28) && got * 8 > 32;
    return HVC_OK;
}
} // namespace hvc

extern "C" {

int hvc_jpeg_entropy_encode(const hvc_jpeg_info *info, const int16_t *coefs, uint8_t *out, size_t cap, size_t *out_len) try {
    if (!info || !coefs || !out_len || info->n_comp != 3) return HVC_E_INVALID_ARG;
    std::vector<uint8_t> o;
    o.reserve(info->coef_count / 4 + 1024);
    hvc::jpeg_header_bytes(info, o);
    if (hvc_jpeg_encoder_check(info)) return HVC_E_INVALID_ARG;
    EncTable et[2];
    build_enc(et[0], K_DC_LUMA_BITS, K_DC_VALS, K_AC_LUMA_BITS, K_AC_LUMA_VALS);
    build_enc(et[1], K_DC_CHROMA_BITS, K_DC_VALS, K_AC_CHROMA_BITS, K_AC_CHROMA_VALS);
    int dc_pred[3] = {0, 0, 0};
    const hvc_jpeg_component &c0 = info->comp[0];
    const int mbs_wide = c0.decoded_width / (8 * c0.hscale), mbs_high = c0.decoded_height / (8 * c0.vscale);
    int per_mcu = 0;
    for (int i = 0; i < 3; i++) per_mcu += info->comp[i].hscale * info->comp[i].vscale;
    // one MCU row of bits without stuffing: a block is at most 64 fields of 27 bits (216 bytes)
    std::vector<uint8_t> scratch((size_t)mbs_wide * (size_t)per_mcu * 216 + 64);
    BitWriter bw(scratch.data());
    for (int my = 0; my < mbs_high; my++) {
        for (int mx = 0; mx < mbs_wide; mx++)
            for (int i = 0; i < 3; i++) {
                const hvc_jpeg_component &c = info->comp[i];
                const hvc_component &L = info->layout[i];
                const EncTable &t = et[c.dc_table];
                for (int sy = 0; sy < c.vscale; sy++)
                    for (int sx = 0; sx < c.hscale; sx++) {
                        const int bx = mx * c.hscale + sx, by = my * c.vscale + sy;
                        const int16_t *q = coefs + L.coef_offset + ((size_t)by * L.blocks_w + bx) * 64;
                        // DC: difference to the predictor (encoder.ml:138-140), size + magnitude (:155-160)
                        const int diff = q[0] - dc_pred[i];
                        dc_pred[i] = q[0];
                        int size = bit_size(diff);
                        if (size > 11) return HVC_E_RANGE; // no code in the default DC tables
                        // code and magnitude bits leave as one field (<= 16 + 11 bits); a negative value's bits are those
                        // of value - 1 (encoder.ml:155-160)
                        bw.put(t.dcf[size].hi | ((unsigned)(diff + (diff >> 31)) & t.dcf[size].mask), t.dcf[size].total);
                        // AC: runs of zeros, ZRL for runs >= 16, EOB when the tail is zero (:127-141, 162-187).
                        // The non-zero positions come from one 64-bit mask (SSE2 compares), so the loop runs
                        // once per coded coefficient with no data-dependent "is it zero" branch.
                        uint64_t nz = nonzero_mask(q) & ~1ull;
                        int prev = 0;
                        while (nz) {
                            const int k = __builtin_ctzll(nz);
                            nz &= nz - 1;
                            int run = k - prev - 1;
                            prev = k;
                            while (run >= 16) { bw.put(t.acf[0xf0].hi, t.acf[0xf0].total); run -= 16; }
                            const int v = q[k];
                            size = bit_size(v);
                            if (size > 10) return HVC_E_RANGE; // no code in the default AC tables
                            const EncField e = t.acf[(run << 4) | size];
                            bw.put(e.hi | ((unsigned)(v + (v >> 31)) & e.mask), e.total);
                        }
                        if (prev != 63) bw.put(t.acf[0].hi, t.acf[0].total);
                    }
            }
        if (my + 1 == mbs_high) bw.pad_with_1s();
        // the row's complete bytes leave; the byte in progress stays in the accumulator
        append_stuffed(o, scratch.data(), (size_t)(bw.cur - scratch.data()));
        bw.cur = scratch.data();
    }
    put_marker(o, 0xd9);
    *out_len = o.size();
    if (!out || o.size() > cap) return HVC_E_INVALID_ARG;
    std::memcpy(out, o.data(), o.size());
    return HVC_OK;
} HVC_ABI_CATCH

// Ocompare.max_difference / total_difference / square_error (tools/src/ocompare.ml:6-47) of two
// equally sized planes given as n bytes each; the float metrics (mean_difference, mean_square_error,
// psnr, :30-59) are one division / log10 on top and stay with the caller.
int hvc_compare_planes(const uint8_t *a, const uint8_t *b, size_t n, int *max_difference, uint64_t *total_difference,
                       uint64_t *square_error) try {
    if ((!a || !b) && n) return HVC_E_INVALID_ARG;
    int mx = 0;
    uint64_t tot = 0, se = 0;
    for (size_t i = 0; i < n; i++) {
        const int d = a[i] > b[i] ? a[i] - b[i] : b[i] - a[i];
        mx = d > mx ? d : mx;
        tot += (uint64_t)d;
        se += (uint64_t)(d * d);
    }
    if (max_difference) *max_difference = mx;
    if (total_difference) *total_difference = tot;
    if (square_error) *square_error = se;
    return HVC_OK;
} HVC_ABI_CATCH

} // extern "C"
