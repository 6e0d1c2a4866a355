// hvc_hdec.h -- parameter blocks of the GPU Huffman DEcoder (internal).
#ifndef HVC_HDEC_H
#define HVC_HDEC_H

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <vector>

struct hvc_jpeg_info;

namespace hvc {

#define HVC_HD_SUBTABLES 8 /* 10-bit prefixes that continue into longer codes and have a 64-entry sub-table, per table */
/* A table with MORE such prefixes (codes of 11..16 bits under more than 8 different 10-bit prefixes: the optimised
 * tables of content with a large alphabet) keeps sub-tables for the first eight -- canonical order: the shortest, most
 * frequent long codes -- and marks the other prefixes HVC_HD_OVF: their symbols are found by the canonical search of
 * ITU-T T.81 F.2.2.3 (per length L: code - mincode[L] < count[L]) in a small record in DEVICE memory (HdOvf*), which
 * any table fits whatever its shape.  Round 2 sent such a file's whole chunk to the host reader (94 -> 8 Gpixel/s). */
#define HVC_HD_OVF 0x3ffu

// One Huffman table as the GPU decoder reads it (built on the host from the DHT segment the model's
// find_huffman_table picks, decoder.ml:238-259; canonical code assignment of tables.ml:27-45).
struct HdTable {
    // (length << 8) | value for codes of <= 10 bits, indexed by the next 10 bits; 0 = no code;
    // 0x8000 | n = the code is longer: entry sub[n * 64 + next 6 bits] has it (two-level table)
    uint16_t fast[1024];
    uint16_t sub[HVC_HD_SUBTABLES * 64];
};
// The codes of 11..16 bits of one table in canonical form (tables.ml:27-45 assigns them in this order): for length
// 11 + i the first code, how many there are and where their symbols start in vals[] -- empty (count all zero) for a
// table that needs no overflow.
struct HdOvfRaw {
    uint16_t mincode[6], count[6], valptr[6];
    uint8_t vals[256]; // the table's symbol values, canonical order (all of them: valptr indexes this)
    uint8_t lens[256]; // ... and their code lengths
    uint16_t used, pad; // used != 0: some prefix of this table is marked HVC_HD_OVF
};
struct HdTables {
    HdTable dc[3], ac[3]; // per scan component (the GPU path takes at most three)
    HdOvfRaw ovf_dc[3], ovf_ac[3];
};
// ... and as the walks read it: the entry of the symbol at canonical position k in the two entry formats
struct HdOvf {
    uint16_t mincode[6], count[6], valptr[6], pad[2];
    uint16_t spec[256]; // HdSpec's format (synchronisation walk)
    uint16_t val[256];  // val_entry's format (k_hd_write2)
};

// The same tables as the synchronisation walk wants them: it needs no values, only how far a symbol moves the
// bit position and the zig-zag index.  Entry: bits 0-5 bits consumed (code + magnitude; 1 for "no code": the
// walk steps one bit), bits 6-12 index advance (run + 1; 1 for a DC symbol; 0 where the index stays; 64 for
// EOB, so that "index >= 64" is the one end-of-block test); in the first level "0 bits consumed" = the code is
// longer than 10 bits, and bits 6-15 then hold the number of the sub-table that has it.
// Components that share their tables share a slot; frames with three different table sets keep to k_hd_round.
struct HdSpec {
    uint16_t t[2][2][1024 + HVC_HD_SUBTABLES * 64]; // [slot][0 = DC, 1 = AC]
};
struct HdSpecOvf { // behind the HdSpec in device memory: the overflow records of the same four tables (never in LDS)
    HdOvf o[2][2];
};
// slot of every component, or false when the frame uses more than two different (DC, AC) table pairs
bool make_spec(const HdTables &t, int n_comp, HdSpec &out, unsigned char slot_of_comp[4], unsigned char slot_rep[2],
               HdSpecOvf *ovf = nullptr);
bool tables_use_overflow(const HdTables &t, int n_comp);

// PER-FRAME tables ("PF mode"): a batch in which the files carry different Huffman tables -- optimised per file, as
// libjpeg -optimize, cameras and most web encoders write them -- or a frame with three different table sets.  One
// record per table set in DEVICE memory (no room for them all in LDS), per COMPONENT rather than per slot; a frame
// points at its record through HdParams::tabset_of.  The walks then read their tables through the L1 / L2 instead of
// LDS: slower per symbol, but the batch stays on the GPU instead of dropping to the host reader.
#define HVC_HD_SPEC_T (1024 + HVC_HD_SUBTABLES * 64)
struct HdFrameTabs {
    uint16_t spec[3][2][HVC_HD_SPEC_T]; // [component][0 = DC, 1 = AC] in HdSpec's entry format (synchronisation walk)
    uint16_t val[3][2][HVC_HD_SPEC_T];  // the same tables in the write pass's format (k_hd_write2: see val_entry)
    // bit 0: the third component reads the second one's tables (or there is none): the frame's tables are the first
    // 12 KB of each form, and a workgroup whose subsequences all belong to this frame keeps them in LDS
    unsigned flags, pad[3];
    HdOvf ovf[3][2];                    // overflow records [component][DC, AC] (read in place, never copied to LDS)
};
void make_frame_tabs(const HdTables &t, int n_comp, HdFrameTabs &out);

#ifndef HVC_HD_SUBSEQ_BITS
#define HVC_HD_SUBSEQ_BITS 1024 /* bits per lane in the synchronisation rounds (a multiple of 128) */
#endif
#define HVC_HD_MAX_MCU_BLOCKS 16
// bytes that must be readable behind the segment buffer (HdParams::ecs): a lane of the write pass that reads its bits
// from global memory looks up to four subsequences past the start of the last one
#define HVC_HD_ECS_SLACK 1024
#define HVC_HD_LIST_N 32 /* >= rounds of k_hd_sync + 2 (+ 2 counters of experiment builds) */
// device bytes of the per-subsequence state for n subsequences (hvc_capi_reader.hip carves HdParams' arrays out of it)
#define HVC_HD_STATE_BYTES(n) ((size_t)(n) * (4 * sizeof(unsigned long long) + 3 * sizeof(unsigned)) + HVC_HD_LIST_N * sizeof(unsigned) + 64)

struct HdComp {
    int h, v, bw;      // sampling factors, plane width in blocks
    int mcu_base;      // first block of the component inside an MCU
    size_t coef_off;   // int16 elements from the frame's coefficient record
};

struct HdParams {
    const uint8_t *ecs;        // the frames' unstuffed entropy-coded segments, each followed by zero padding
    const unsigned *ecs_off;   // [n_frames] byte offset of frame f's segment inside ecs (multiple of 4)
    const unsigned *sub_off;   // [n_frames + 1] first subsequence of every frame
    const unsigned *frame_of;  // [total_sub] frame of every subsequence
    const HdTables *tables;    // device
    const HdSpec *spec;        // device, or null: synchronise with k_hd_round only
    const HdSpecOvf *spec_ovf; // device: the overflow records of spec's tables (valid whenever spec is)
    const HdFrameTabs *ftabs;  // device, PF mode (then spec / tables are not used), or null
    const unsigned *tabset_of; // [n_frames] PF mode: index of the frame's record in ftabs
    unsigned selmask;          // 2 bits per block b of an MCU: which tables it reads -- the HdSpec slot (0 / 1), or in PF
                               // mode its component (0..2)
    unsigned slotmask;         // bit b = table slot (HdSpec) of block b of an MCU
    unsigned char slot_rep[2]; // a component whose tables the slot stands for
    int n_frames;
    unsigned total_sub;
    int n_comp, blocks_per_mcu, mbs_wide, mbs_high;
    unsigned blocks_per_frame; // blocks a frame can have (restart mode: rst_mcus MCUs' worth -- the capacity of dcd's rows)
    // Restart intervals (opt-in, beyond the model: include/hvc_jpeg.h hvc_set_restart_markers): every interval of a file is a
    // FRAME of its own to the reader -- its bytes start on a subsequence boundary, its first subsequence has the true start
    // (block 0 of an MCU, DC predictors at zero), nothing synchronises or sums across its ends.  Frame f = interval
    // f % rst_ipf of file f / rst_ipf: its blocks are the MCUs [k rst_mcus, (k + 1) rst_mcus) of the file's scan (the last
    // interval of a file may have fewer) and land in the FILE's coefficient record.  rst_ipf <= 1: off, frame = file.
    unsigned rst_mcus, rst_ipf;
    HdComp comp[4];
    unsigned char b2comp[HVC_HD_MAX_MCU_BLOCKS];
    int16_t *coefs;
    size_t coef_fs;
    // per subsequence: state = bit position | k << 32 | block-in-MCU << 40
    unsigned long long *start_used, *exit_a, *exit_b;
    // k_hd_sync only: second per-round exit buffer (exit_b is the first), the two work lists, list lengths per round
    unsigned long long *exit_c;
    unsigned *list0, *list1, *list_n; // [total_sub], [total_sub], [HVC_HD_LIST_N]
    // PF mode with more than a handful of files: one work list per FILE (its entries sit at list0/1 + the sub_off of its first frame,
    // their number per round in list_fn[round * n_frames + f]), so that a workgroup's entries are all of one frame and
    // its tables can go to LDS.  list_fn: [HVC_HD_LIST_N * n_frames], or null (then the batch-wide lists are used).
    unsigned *list_fn;
    unsigned max_frame_sub;    // the largest number of subsequences any FILE has (the launches' grid; restart intervals: all of a file's frames share a list)
    int16_t *dcd;              // [n_frames * blocks_per_frame] DC differences in scan order (k_hd_write2 -> k_hd_dc), or null
    int16_t *dc_plane;         // null: k_hd_dc puts the DC values into the records.  Otherwise into this compact array,
    size_t dc_fs;              //   dc_plane[frame * dc_fs + (block's coefficient offset in the frame record) / 64] -- what
                               //   DecodeParams::dc_plane reads; the records then keep the DC DIFFERENCE in coefficient 0
    unsigned *nblk;            // blocks completed inside the subsequence, then (scan) index of its first block
    unsigned *frame_blocks;    // [n_frames] blocks found in the whole segment
    unsigned *changed;         // [1]
    unsigned *status;          // [1] bit 0: invalid code / index out of range / DC category > 16 inside the coded blocks,
                               //     bit 1: DC outside int16, bit 2: fewer blocks than the frame needs (truncated stream)
};

hipError_t launch_hd_round(const HdParams &P, int round, hipStream_t s);
#ifdef HVC_HD_STATS
void hd_stats_read(unsigned long long out[4]); // experiments (tools/exp_hd_stats.py)
#endif
hipError_t launch_hd_frame_of(const HdParams &P, hipStream_t s); // fills P.frame_of from P.sub_off (callers that do not upload it)
hipError_t launch_hd_finish(const HdParams &P, int rounds_done, hipStream_t s); // count scan, write pass, DC pass
bool hd_write2_fits(const HdParams &P); // the fast write pass can address these records (PF mode needs it)
inline unsigned hd_files(const HdParams &P) { return P.rst_ipf > 1 ? ((unsigned)P.n_frames + P.rst_ipf - 1) / P.rst_ipf : (unsigned)P.n_frames; }

// hvc_entropy.cpp: header parse + table preparation + unstuffing for one file
// returns HVC_OK, or an hvc_status the host decoder would also return at this stage;
// gpu_ok = false when the stream needs the host decoder (tables that are no prefix code, > 16 blocks per MCU)
int prepare_gpu_decode(const uint8_t *jpeg, size_t n, const ::hvc_jpeg_info *info, HdTables &t, std::vector<uint8_t> &ecs,
                       bool &gpu_ok);
// the same, unstuffing straight into dst[0, cap) (a pinned ring slot); a segment that does not fit clears gpu_ok.
// units (restart intervals honoured on this thread, the file's DRI = units->interval and its scan holds units->ipf > 1 of
// them): interval k goes to dst + off[k], len[k] bytes and zeros behind them to the end of its slot, hd_unit_slot(len[k]);
// *ecs_size = the bytes of all intervals.  A file with restart intervals and no `units` (or other numbers) clears gpu_ok.
struct RstUnits {
    unsigned interval, ipf;
    unsigned *off, *len; // [ipf]
};
// An interval's place in the segment buffer: its subsequences -- the last one filled up with zeros -- and 16 bytes of
// overshoot; a multiple of 16.  (A FILE's segment gets a whole subsequence of zeros more, so that a stream that ends a few
// symbols early still finds its zeros on the GPU; per interval that subsequence -- a thousand zero bits are five hundred
// of the shortest code: the slowest walk of its wavefront, in every round -- cost a third of the reader's time, and it is
// not needed: a block the frame must have and that does not END inside the frame's own subsequences raises the
// truncation flag (k_hd_scan), which sends the file to the host reader.)
inline size_t hd_unit_subs(size_t len) { return len ? (len + HVC_HD_SUBSEQ_BITS / 8 - 1) / (HVC_HD_SUBSEQ_BITS / 8) : 1; }
inline size_t hd_unit_slot(size_t len) { return hd_unit_subs(len) * (HVC_HD_SUBSEQ_BITS / 8) + 16; }
int prepare_gpu_decode_to(const uint8_t *jpeg, size_t n, const ::hvc_jpeg_info *info, HdTables &t, uint8_t *dst, size_t cap,
                          size_t *ecs_size, bool &gpu_ok, RstUnits *units = nullptr);
unsigned restart_interval_of(const uint8_t *jpeg, size_t n); // the file's DRI (0: none, or no header that parses)


// A block whose absolute DC (decoder.ml:143, a 63-bit sum in the model) does not fit the int16 coefficient record:
// the record holds the value saturated to +-32767, this entry the true one.  block = the block's coefficient offset
// inside the frame record / 64.
struct WideDc {
    uint32_t block;
    long long dc;
};
// hvc_jpeg_entropy_decode that goes on where that one answers HVC_E_RANGE: such blocks are listed in `wide` (the
// file-level decode entry points then recompute them in int64 with their true DC, as the model does)
// restart intervals honoured by the host reader on this thread (opt-in, beyond the model: hvc_entropy.cpp)
extern thread_local bool tl_honour_restart;
struct RestartScope {
    bool prev;
    explicit RestartScope(bool on) : prev(tl_honour_restart) { tl_honour_restart = on; }
    ~RestartScope() { tl_honour_restart = prev; }
};
int entropy_decode_wide(const uint8_t *data, size_t n, const ::hvc_jpeg_info *info, int16_t *coefs, std::vector<WideDc> &wide);
// the same for two files at once on one thread, their symbols decoded in turn (two dependency chains for the core to
// overlap); st[i] = what entropy_decode_wide would have returned for file i
void entropy_decode_wide2(const uint8_t *const data[2], const size_t n[2], const ::hvc_jpeg_info *const info[2], int16_t *const coefs[2],
                          std::vector<WideDc> *const wide[2], int st[2]);

} // namespace hvc
#endif
