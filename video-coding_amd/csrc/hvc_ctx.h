// hvc_ctx.h -- what the translation units of the C ABI share (internal; nothing here is exported): the context, the
// helpers every entry point uses, and the few functions one part of the ABI calls in another.
//   hvc_capi.hip         context, streams, timers, memory; the block stage (hvc_decode_frames, hvc_encode_frames, ...)
//   hvc_capi_jpeg.hip    files: one at a time (hvc_jpeg_decode, hvc_jpeg_encode) and the batch pipeline with the host reader
//   hvc_capi_reader.hip  the GPU Huffman reader's entry point and the batch pipeline built on it
//   hvc_capi_files.hip   the GPU Huffman coder's entry point and the batch pipelines that write files
//   hvc_capi_async.hip   pinned host memory and the slots of the asynchronous seam (hvc_decode_frames_submit / hvc_wait)
#ifndef HVC_CTX_H
#define HVC_CTX_H

#include <hip/hip_runtime.h>
#include <pthread.h>
#include <sched.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <thread>
#include <deque>
#include <functional>
#include <vector>

#include "../../include/hvc_jpeg.h"
#include "hvc_hdec.h"
#include "hvc_huff.h"
#include "hvc_kernels.h"
#include "hvc_pool.h"

#define HVC_PROF_RING 64
#define HVC_FIX_WORDS 8 /* d_fix_count: [0] [1] counters, [2..3] the 64-bit total, [4] [5] the fused path's luma counters */

struct hvc_ctx {
    int device = -1;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // ring of event pairs around the dominant kernel of the last HVC_PROF_RING profiled calls
    hipEvent_t k0[HVC_PROF_RING] = {}, k1[HVC_PROF_RING] = {};
    unsigned long long k_calls = 0;
    bool profiling = false;
    bool honour_restart = false; // hvc_set_restart_markers: restart intervals honoured by the file-level entry points (an extension)
    int decode_kernel = 0; // hvc_set_decode_kernel: 0 packed (default), 1 unpacked int32, 2 int64 for every block, 3 q16
    unsigned *d_fix_count = nullptr; // two counters, used alternately (see k_decode_wide); behind them (+ 8 bytes) the 64-bit
                                     // total of the last call's fix-up blocks over all its launches (hvc_last_wide_blocks)
    // [4], [5]: a second pair of counters, for the luma planes of the fused 4:4:4 path when they run through
    // k_decode_packed beside (or before) the chroma tiles' kernel, which uses the first pair
    int fix_phase_l = 0;
    hipStream_t side_stream = nullptr; // ... and the stream that kernel runs on in the side-by-side form, with its fork / join events
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool wide_total_started = false; // the current call has enqueued a launch that stores (rather than adds to) that total
    long long wide_host = -1;        // >= 0: the last call sent every block through the int64 kernel (no list): this many
    int fix_phase = 0;               // index of the counter the NEXT decode call appends to
    int fix_last = 0;                // index of the counter the last decode call used
    unsigned *d_fix_list = nullptr;
    size_t fix_cap = 0; // entries
    void *d_in = nullptr, *d_out = nullptr, *d_sums = nullptr, *d_aux = nullptr, *d_aux2 = nullptr;
    size_t in_cap = 0, out_cap = 0, sums_cap = 0, aux_cap = 0, aux2_cap = 0;
    int last_hip = 0;
    // hvc_jpeg_decode_batch: copy stream + ring of pinned host / device coefficient chunks
    static constexpr int RING = 3;
    hipStream_t copy_stream = nullptr, down_stream = nullptr;
    hipStream_t rd_stream[3] = {}; // hvc_jpeg_decode_batch_gpu: the Huffman reader of even / odd chunks
    hipEvent_t ev_rd[3] = {};      // ... its "records complete" per ring slot (RING entries)
    void *h_ring[RING] = {}, *d_ring[RING] = {}, *d_oring[RING] = {};
    size_t ring_bytes = 0, oring_bytes = 0;
    hipEvent_t ev_h2d[RING] = {}, ev_kern[RING] = {}, ev_t[4] = {};
    // hvc_jpeg_encode_batch: pinned / device rings of padded pixel chunks (in) and coefficient chunks (out)
    void *eh_in[RING] = {}, *ed_in[RING] = {}, *eh_out[RING] = {}, *ed_out[RING] = {};
    size_t e_in_bytes = 0, e_out_bytes = 0;
    hipEvent_t ev_up[RING] = {}, ev_down[RING] = {}, ev_et[RING][3] = {}, ev_gpu[RING] = {};
    void *ed_seg[RING] = {}, *ed_off[RING] = {}, *eh_off[RING] = {}; // hvc_jpeg_encode_batch_gpu: packed segments + offsets
    size_t e_seg_bytes = 0, e_off_bytes = 0;
    // hvc_jpeg_decode_batch_gpu: pinned / device rings of unstuffed segments and their index arrays
    void *gp_h_ecs[RING] = {}, *gp_d_ecs[RING] = {}, *gp_h_meta[RING] = {}, *gp_d_meta[RING] = {};
    void *gp_h_ftabs[RING] = {}, *gp_d_ftabs[RING] = {}; // ... and of per-frame Huffman tables (hvc::HdFrameTabs, PF mode)
    size_t gp_ecs_bytes = 0, gp_meta_bytes = 0, gp_ftabs_bytes = 0;
    // GPU Huffman decoder (hvc_jpeg_entropy_decode_gpu): device scratch, grown on demand
    void *gd_ecs = nullptr, *gd_meta = nullptr, *gd_state = nullptr, *gd_tables = nullptr, *gd_coefs = nullptr, *gd_dcd = nullptr;
    void *gd_h_ecs = nullptr; // pinned: the batch's unstuffed segments on their way to gd_ecs
    size_t gd_h_ecs_cap = 0;
    void *gd_fcnt = nullptr;  // PF mode: per-frame list lengths per round (hvc::HdParams::list_fn)
    size_t gd_fcnt_cap = 0;
    void *gd_ftabs = nullptr; // per-frame tables of hvc_jpeg_entropy_decode_gpu (PF mode)
    size_t gd_ftabs_cap = 0;
    void *gd_dcv = nullptr;   // batch pipeline: the blocks' DC values as a compact array (hvc::DecodeParams::dc_plane)
    size_t gd_dcv_cap = 0;
    void *d_dcfix = nullptr;  // blocks with a DC beyond int16 (hvc::WideDc): ids, true DCs, count
    size_t dcfix_cap = 0;
    // hvc_set_host_cpus: the CPUs the batch pipelines' host threads may run on (empty = no restriction)
    bool have_cpus = false;
    cpu_set_t cpus;
    char cpulist[256] = "";
    bool have_default_cpus = false; // the process's own mask when the context was created: what the pool's threads go
    cpu_set_t default_cpus;         // back to when a restriction is lifted (they outlive the call that pinned them)
    hvc::WorkerPool pool;           // the batch pipelines' host threads (hvc_pool.h): persistent, joined in hvc_destroy
    hvc::HdTables *gd_tables_host = nullptr; // what gd_tables holds (value tables; the HdSpec behind them follows from these)
    bool gd_tables_valid = false;
    int gd_tables_ncomp = 0;
    size_t gd_ecs_cap = 0, gd_meta_cap = 0, gd_state_cap = 0, gd_tables_cap = 0, gd_coefs_cap = 0, gd_dcd_cap = 0;
    // GPU Huffman coder (hvc_huffman_encode_frames): tables + scratch, grown on demand
    unsigned *hd_tables = nullptr;
    void *hd_lens = nullptr, *hd_meta = nullptr, *hd_bitbuf = nullptr, *hd_ff = nullptr, *hd_out = nullptr;
    size_t hd_lens_cap = 0, hd_meta_cap = 0, hd_bitbuf_cap = 0, hd_ff_cap = 0, hd_out_cap = 0;
    // The asynchronous seam (hvc_capi_async.hip): one batch in flight per slot.  Uploads run on copy_stream, the block stage
    // on `stream`, downloads on down_stream; a slot's device buffers are its own (grown while the slot is free).
    struct Slot {
        void *d_in = nullptr, *d_out = nullptr;
        size_t in_cap = 0, out_cap = 0;
        // up0 / up1 around the upload, k0 / k1 around the block stage, dn0 / dn1 around the download; `done` = the last of them
        hipEvent_t up0 = nullptr, up1 = nullptr, k0 = nullptr, k1 = nullptr, dn0 = nullptr, dn1 = nullptr;
        bool busy = false, has_down = false, timed = false;
        unsigned long long h2d_bytes = 0, d2h_bytes = 0;
    } slots[HVC_SLOTS];
};

// pinned rings the host only ever writes (unstuffed segments, padded raw frames) and the copy engine reads
#ifndef HVC_UPLOAD_RING_FLAGS
#define HVC_UPLOAD_RING_FLAGS hipHostMallocDefault
#endif

struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() {
        int cur = -1;
        if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
    }
};

inline int fail_hip(hvc_ctx *c, hipError_t e) {
    c->last_hip = (int)e;
    return HVC_E_HIP;
}
#define HIPCHK(c, call)                                 \
    do {                                                \
        hipError_t e_ = (call);                         \
        if (e_ != hipSuccess) return fail_hip((c), e_); \
    } while (0)

// Every host thread a batch pipeline starts calls this first (hvc_set_host_cpus); false = the restriction could not
// be applied (the batch call then fails rather than run somewhere it was told not to).
bool pin_to_ctx_cpus(const hvc_ctx *c);

// `n` pool threads for a pipeline call (+ `extra` for its downloader): HVC_OK, or HVC_E_SYSTEM when the system refuses one
inline int pool_ready(hvc_ctx *c, int n, int extra = 0) { return c->pool.ensure(n + extra); }

// Linux cpulist format ("0-15,32-47") -> cpu_set_t; false on a syntax error, an empty set or a CPU beyond CPU_SETSIZE
bool parse_cpulist(const char *s, cpu_set_t &set);

// device scratch *p of at least `need` bytes (grown with a quarter to spare; the stream is drained before the old one goes)
int grow(hvc_ctx *c, void **p, size_t *cap, size_t need);

// The two fix-up counters alternate between launches: a launch appends to counter fix_phase and its wide kernel
// clears the other one for the launch after it (no memset node).  The roles change hands only once a launch has
// been enqueued: a call that fails before or while launching leaves fix_phase where it was and, if anything may
// have reached the stream, both counters are cleared -- the next call must never find a stale count (its wide kernel
// would re-process old list entries under the new geometry).
template <class Params>
inline void fix_assign(const hvc_ctx *c, Params &P) {
    P.fix_count = c->d_fix_count + c->fix_phase;
    P.fix_count_next = c->d_fix_count + (c->fix_phase ^ 1);
    P.fix_list = c->d_fix_list;
    P.wide_total = reinterpret_cast<unsigned long long *>(c->d_fix_count + 2);
    P.wide_first = c->wide_total_started ? 0 : 1;
}
inline void fix_commit(hvc_ctx *c) {
    c->fix_last = c->fix_phase;
    c->fix_phase ^= 1;
    c->wide_total_started = true; // the call's next launches add to the total
}
// at the start of every decode call: its first launch starts the total over
inline void wide_total_begin(hvc_ctx *c) {
    c->wide_total_started = false;
    c->wide_host = -1;
}
inline void fix_reset(hvc_ctx *c) { // after a failed launch: all counters to zero, in stream order
    (void)hipMemsetAsync(c->d_fix_count, 0, HVC_FIX_WORDS * sizeof(unsigned), c->stream); // (and the total behind the first pair)
}

// Launches longer than about 3 ms lose 2-3 % against back-to-back shorter ones (measured on MI355X: 1080p batches of
// 2048 / 4096 frames per launch run at 71.9 / 71.6 % of the HBM peak, 1024-frame launches -- even 1900 of them back to
// back over 3 s, or sixteen of them over a 154 GB resident set -- at 74.4 %; the counters show a lower clock and more
// DRAM read-credit stalls late in a long launch, not TLB misses: DESIGN.md section 5).  So a device-memory batch is cut
// into launches of at most this many algorithmic bytes (192 B per block); HVC_LAUNCH_BYTES overrides (experiments).
size_t launch_bytes_limit();
// The fused 4:4:4 path's block stage (decode_frames_yuv444_impl): 0 = one kernel for luma and chroma tiles, 1 = the luma
// planes through k_decode_packed, then the chroma tiles, 2 = the two side by side on two streams.  HVC_444_MODE
// overrides the default (A/B measurements).
#ifndef HVC_444_MODE_DEFAULT
#define HVC_444_MODE_DEFAULT 0
#endif
int fused444_mode();
// frames per launch for a batch of n_frames frames of blocks_per_frame blocks: equal parts, each within the limit
int frames_per_launch(int n_frames, unsigned long long blocks_per_frame);

// Geometry of one call -> CompK[]; shared by decode and encode.
struct Layout {
    hvc::CompK comp[HVC_MAX_COMP];
    int n_comp = 0, tiles_per_frame = 0;
    size_t coef_span = 0;  // elements covered by one frame record
    size_t pixel_span = 0; // bytes covered by one frame record
    unsigned long long blocks_per_frame = 0;
};
int make_layout(const hvc_component *comps, int n_comp, int n_qtabs, Layout &L);
int check_qtabs(const uint16_t *qtabs, int n_qtabs, bool divides); // divides: the encoder (an entry of zero is HVC_E_RANGE)

// A component without a block (blocks_w or blocks_h of zero: the model's empty Plane.t) has no part in the block stage: the
// decoding entry points drop such components from the list they work on; *n_kept == 0: nothing to decode at all.
int drop_empty_components(const hvc_component *comps, int n_comp, hvc_component *kept, int *n_kept);

// What comes back to a host caller is what the kernels wrote, never the padding between planes.  Planes (pixel planes:
// decode; coefficient planes: encode) that follow one another tightly -- the usual record -- are ONE stretch per frame:
// [first, first + len) in bytes from the frame record; len == 0: they are not, every plane is copied by itself.
struct RecordRun {
    size_t first = 0, len = 0;
};
RecordRun pixel_run(const hvc_component *comps, int n_comp);
RecordRun coef_run(const hvc_component *comps, int n_comp); // (bytes, like pixel_run)
// frames [f0, f0 + cnt) of a batch laid out alike on the device (d_base) and on the host (h_base), frame stride `fs` bytes,
// device -> host on stream `st`: one copy (stretches that touch), one 2D copy, or one 2D copy per plane
hipError_t download_pixels(const hvc_component *comps, int n_comp, const RecordRun &run, int f0, int cnt, size_t fs,
                           const uint8_t *d_base, uint8_t *h_base, hipStream_t st);
hipError_t download_coefs(const hvc_component *comps, int n_comp, const RecordRun &run, int f0, int cnt, size_t fs_bytes,
                          const uint8_t *d_base, uint8_t *h_base, hipStream_t st);

// The batch pipelines' orchestrating thread waits most of the call (an upload's end, a chunk's kernels).
// hipEventSynchronize spins -- also on an event created with hipEventBlockingSync, on this ROCm (measured: CPU time =
// wall time) -- and on a box whose processes own a fixed share of CPU time (16 CPUs for one GPU here) a spinning thread
// takes its CPU from the workers that are the bound of the pipeline.  So: poll and sleep, 20 us at first, 200 us from
// the tenth poll on (the waits are milliseconds long).  HVC_EVENT_SPIN=1: hipEventSynchronize (A/B).
hipError_t wait_event(hipEvent_t e);

// Host buffers, large batches: four parts; while part k + 1 is uploaded (c->stream), part k runs through the kernels
// (c->stream) and is downloaded (a second thread on c->down_stream: copies to and from pageable memory hold
// their caller), so the link carries both directions at once.  up(f0, cnt) / run(k, f0, cnt) enqueue on c->stream,
// down(f0, cnt, stream) on the stream it is given.
template <class Up, class Run, class Down>
inline int overlapped_parts(hvc_ctx *c, int n_frames, Up up, Run run, Down down) {
    constexpr int K = 4;
    if (!c->down_stream) HIPCHK(c, hipStreamCreateWithFlags(&c->down_stream, hipStreamNonBlocking));
    for (int i = 0; i < K; i++)
        if (!c->ev_t[i]) HIPCHK(c, hipEventCreate(&c->ev_t[i]));
    std::atomic<int> launched{0}, herr{0};
    auto part = [&](int k, int &f0, int &cnt) {
        f0 = (int)((long long)n_frames * k / K);
        cnt = (int)((long long)n_frames * (k + 1) / K) - f0;
    };
    int pr = pool_ready(c, 1);
    if (pr) return pr;
    hvc::PoolScope scope(c->pool, [&] { if (launched.load() < K) herr.store(herr.load() ? herr.load() : (int)hipErrorUnknown); });
    pr = c->pool.submit([&] {
        (void)pin_to_ctx_cpus(c);
        if (hipSetDevice(c->device) != hipSuccess) { herr.store((int)hipErrorInvalidDevice); return; }
        for (int k = 0; k < K; k++) {
            while (launched.load(std::memory_order_acquire) <= k && !herr.load()) std::this_thread::yield();
            if (herr.load()) return;
            int f0, cnt;
            part(k, f0, cnt);
            hipError_t e = hipStreamWaitEvent(c->down_stream, c->ev_t[k], 0);
            if (e == hipSuccess) e = down(f0, cnt, c->down_stream);
            if (e == hipSuccess) e = hipStreamSynchronize(c->down_stream);
            if (e != hipSuccess) { herr.store((int)e); return; }
        }
    }, 1);
    if (pr) return pr;
    hipError_t e = hipSuccess;
    for (int k = 0; k < K && e == hipSuccess && !herr.load(); k++) {
        int f0, cnt;
        part(k, f0, cnt);
        e = up(f0, cnt);
        if (e == hipSuccess) e = run(k, f0, cnt);
        if (e == hipSuccess) e = hipEventRecord(c->ev_t[k], c->stream);
        if (e == hipSuccess) launched.store(k + 1, std::memory_order_release);
    }
    if (e != hipSuccess) herr.store((int)e);
    pr = scope.finish();
    (void)hipStreamSynchronize(c->stream);
    if (pr) return pr;
    if (herr.load()) return fail_hip(c, (hipError_t)herr.load());
    return HVC_OK;
}

// ---------------------------------------------------------------------------
// What one part of the ABI calls in another

// A block of a batch whose true DC does not fit the int16 record (hvc::WideDc of frame `frame` of the batch): after
// the batch's launches it is recomputed in int64 with that DC -- what the model's 63-bit arithmetic gives.
struct WideFix {
    int frame;
    uint32_t block;
    long long dc;
};

// hvc_capi.hip: the block stage behind hvc_decode_frames / hvc_decode_frames_yuv444.
// dc_plane (device memory calls only, default kernels only): see hvc::DecodeParams::dc_plane
// wide (device memory calls only): blocks to recompute with their true DC once the launches are enqueued
int decode_frames_impl(hvc_ctx *c, const int16_t *coefs, size_t coef_fs, const uint16_t *qtabs, int n_qtabs,
                       const hvc_component *comps, int n_comp, int n_frames, uint8_t *pixels, size_t pixel_fs, int where,
                       const int16_t *dc_plane, size_t dc_fs, const std::vector<WideFix> *wide = nullptr);
int decode_frames_yuv444_impl(hvc_ctx *c, const int16_t *coefs, size_t coef_fs, const uint16_t *qtabs, int n_qtabs,
                              const hvc_component *comps, int n_comp, int n_frames, int width, int height, uint8_t *frames,
                              size_t frame_stride, int where, const int16_t *dc_plane, size_t dc_fs,
                              const std::vector<WideFix> *wide = nullptr);

// after_reader (optional): called once the reader's launches are enqueued and BEFORE its verdict is known -- the caller
// enqueues what consumes the records (block stage, download) on the same stream, so that one call costs one host
// synchronisation instead of two; *speculated tells whether what it enqueued ran on valid records.
struct AfterReader {
    std::function<int()> enqueue; // an hvc_status
    bool speculated = false;      // out: enqueue() ran, and behind a reader run whose verdict was good
    // The consumer is the block stage: the reader's DC pass then writes the DC values to this compact array
    // (hvc::DecodeParams::dc_plane, one per block of the frame record) instead of 2 bytes into each 128-byte record --
    // a partial-line write apiece, 35 of a single file's 430 us -- and the records keep the DC difference.
    int16_t *dc_plane = nullptr;
    size_t dc_fs = 0;
};
// hvc_capi_reader.hip: Huffman decoding on the GPU (hvc_hdec.hip) of a batch of files that share a geometry.  HVC_OK with
// *used_gpu = 1 when the coefficient records at d_coefs are complete; HVC_OK with *used_gpu = 0 when a stream needs the
// host reader (nothing usable was written); or the error the host reader would report while parsing headers.
int gpu_entropy_decode(hvc_ctx *c, const uint8_t *const *jpegs, const size_t *sizes, int n_frames, const hvc_jpeg_info &info0,
                       int16_t *d_coefs, size_t coef_fs, int *used_gpu, AfterReader *after = nullptr);

// hvc_capi_jpeg.hip: the batch pipeline with the host reader (hvc_jpeg_decode_batch / _yuv444) -- also where the
// pipeline with the GPU reader sends a batch it cannot take
int decode_batch_impl(hvc_ctx *c, const uint8_t *const *jpegs, const size_t *sizes, int n_frames, int threads,
                      int frames_per_chunk, uint8_t *pixels, size_t pixel_fs, int where, hvc_batch_stats *stats, bool yuv444);
inline bool is_420_scan(const hvc_jpeg_info &info) { // Y 2x2, Cb / Cr 1x1 (frame.ml:42-61)
    return info.n_comp == 3 && info.comp[0].hscale == 2 && info.comp[0].vscale == 2 && info.comp[1].hscale == 1 &&
           info.comp[1].vscale == 1 && info.comp[2].hscale == 1 && info.comp[2].vscale == 1;
}

// hvc_capi_files.hip: geometry + scratch + tables of one GPU Huffman coder call (hvc_huff.hip)
int huffman_prepare(hvc_ctx *c, const hvc_jpeg_info *info, const int16_t *d_coefs, size_t coef_fs, int n_frames, uint8_t *d_out,
                    size_t out_cap, unsigned long long *d_offsets, hvc::HuffParams &P);

#endif
