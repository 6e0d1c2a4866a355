// hvc_capi.hip -- the C ABI of include/hvc_jpeg.h over the gfx950 kernels.
// Host-side plumbing only: argument checking, geometry -> kernel parameter
// blocks, staging for host-memory callers, streams and events.
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
};

namespace {

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

int fail_hip(hvc_ctx *c, hipError_t e) {
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
bool pin_to_ctx_cpus(const hvc_ctx *c) {
    if (!c->have_cpus) { // a pool thread may still carry an earlier call's restriction
        if (c->have_default_cpus) (void)pthread_setaffinity_np(pthread_self(), sizeof(cpu_set_t), &c->default_cpus);
        return true;
    }
    return pthread_setaffinity_np(pthread_self(), sizeof(cpu_set_t), &c->cpus) == 0;
}

// `n` pool threads for a pipeline call (+ `extra` for its downloader): HVC_OK, or HVC_E_SYSTEM when the system refuses one
int pool_ready(hvc_ctx *c, int n, int extra = 0) { return c->pool.ensure(n + extra); }

// Linux cpulist format ("0-15,32-47") -> cpu_set_t; false on a syntax error, an empty set or a CPU beyond CPU_SETSIZE
bool parse_cpulist(const char *s, cpu_set_t &set) {
    CPU_ZERO(&set);
    int n = 0;
    while (*s) {
        while (*s == ' ' || *s == '\n') s++;
        if (!*s) break;
        char *end = nullptr;
        const long a = std::strtol(s, &end, 10);
        if (end == s || a < 0) return false;
        long b = a;
        s = end;
        if (*s == '-') {
            b = std::strtol(s + 1, &end, 10);
            if (end == s + 1 || b < a) return false;
            s = end;
        }
        if (b >= CPU_SETSIZE) return false;
        for (long k = a; k <= b; k++) CPU_SET((int)k, &set), n++;
        while (*s == ' ' || *s == '\n') s++;
        if (*s == ',') s++;
        else if (*s) return false;
    }
    return n > 0;
}

int grow(hvc_ctx *c, void **p, size_t *cap, size_t need) {
    if (need <= *cap) return HVC_OK;
    if (*p) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, hipFree(*p));
        *p = nullptr;
        *cap = 0;
    }
    size_t want = need + need / 4 + 4096;
    hipError_t e = hipMalloc(p, want);
    if (e != hipSuccess) {
        *p = nullptr;
        c->last_hip = (int)e;
        return HVC_E_OUT_OF_MEMORY;
    }
    *cap = want;
    return HVC_OK;
}

// The two fix-up counters alternate between launches: a launch appends to counter fix_phase and its wide kernel
// clears the other one for the launch after it (no memset node).  The roles change hands only once a launch has
// been enqueued: a call that fails before or while launching leaves fix_phase where it was and, if anything may
// have reached the stream, both counters are cleared -- the next call must never find a stale count (its wide kernel
// would re-process old list entries under the new geometry).
template <class Params>
static void fix_assign(const hvc_ctx *c, Params &P) {
    P.fix_count = c->d_fix_count + c->fix_phase;
    P.fix_count_next = c->d_fix_count + (c->fix_phase ^ 1);
    P.fix_list = c->d_fix_list;
    P.wide_total = reinterpret_cast<unsigned long long *>(c->d_fix_count + 2);
    P.wide_first = c->wide_total_started ? 0 : 1;
}
static void fix_commit(hvc_ctx *c) {
    c->fix_last = c->fix_phase;
    c->fix_phase ^= 1;
    c->wide_total_started = true; // the call's next launches add to the total
}
// at the start of every decode call: its first launch starts the total over
static void wide_total_begin(hvc_ctx *c) {
    c->wide_total_started = false;
    c->wide_host = -1;
}
static void fix_reset(hvc_ctx *c) { // after a failed launch: all counters to zero, in stream order
    (void)hipMemsetAsync(c->d_fix_count, 0, HVC_FIX_WORDS * sizeof(unsigned), c->stream); // (and the total behind the first pair)
}

// Launches longer than about 3 ms lose 2-3 % against back-to-back shorter ones (measured on MI355X: 1080p batches of
// 2048 / 4096 frames per launch run at 71.9 / 71.6 % of the HBM peak, 1024-frame launches -- even 1900 of them back to
// back over 3 s, or sixteen of them over a 154 GB resident set -- at 74.4 %; the counters show a lower clock and more
// DRAM read-credit stalls late in a long launch, not TLB misses: DESIGN.md section 5).  So a device-memory batch is cut
// into launches of at most this many algorithmic bytes (192 B per block); HVC_LAUNCH_BYTES overrides (experiments).
static size_t launch_bytes_limit() {
    static const size_t v = [] {
        const char *e = std::getenv("HVC_LAUNCH_BYTES");
        const double d = e ? std::atof(e) : 0.0;
        return d >= 1.0 ? (size_t)d : (size_t)10000000000ull;
    }();
    return v;
}
// The fused 4:4:4 path's block stage (decode_frames_yuv444_impl): 0 = one kernel for luma and chroma tiles, 1 = the luma
// planes through k_decode_packed, then the chroma tiles, 2 = the two side by side on two streams.  HVC_444_MODE
// overrides the default (A/B measurements).
#ifndef HVC_444_MODE_DEFAULT
#define HVC_444_MODE_DEFAULT 0
#endif
static int fused444_mode() {
    static const int m = [] {
        const char *e = std::getenv("HVC_444_MODE");
        const int v = e ? std::atoi(e) : HVC_444_MODE_DEFAULT;
        return v < 0 || v > 2 ? 0 : v;
    }();
    return m;
}
// frames per launch for a batch of n_frames frames of blocks_per_frame blocks: equal parts, each within the limit
static int frames_per_launch(int n_frames, unsigned long long blocks_per_frame) {
    const unsigned long long fb = blocks_per_frame * 192ull;
    unsigned long long per = fb ? launch_bytes_limit() / fb : (unsigned long long)n_frames;
    if (per < 1) per = 1;
    if (per >= (unsigned long long)n_frames) return n_frames;
    const unsigned long long parts = ((unsigned long long)n_frames + per - 1) / per;
    return (int)(((unsigned long long)n_frames + parts - 1) / parts);
}

// Geometry of one call -> CompK[]; shared by decode and encode.
struct Layout {
    hvc::CompK comp[HVC_MAX_COMP];
    int n_comp = 0, tiles_per_frame = 0;
    size_t coef_span = 0;  // elements covered by one frame record
    size_t pixel_span = 0; // bytes covered by one frame record
    unsigned long long blocks_per_frame = 0;
};

int make_layout(const hvc_component *comps, int n_comp, int n_qtabs, Layout &L) {
    if (!comps || n_comp < 1 || n_comp > HVC_MAX_COMP) return HVC_E_INVALID_ARG;
    L.n_comp = n_comp;
    int tile = 0;
    for (int i = 0; i < n_comp; i++) {
        const hvc_component &c = comps[i];
        if (c.blocks_w < 1 || c.blocks_h < 1 || c.qtab < 0 || c.qtab >= n_qtabs) return HVC_E_INVALID_ARG;
        if (c.stride < (size_t)c.blocks_w * 8) return HVC_E_INVALID_ARG;
        if ((c.stride & 7) || (c.plane_offset & 7) || (c.coef_offset & 7)) return HVC_E_ALIGNMENT;
        unsigned long long nblk = (unsigned long long)c.blocks_w * (unsigned long long)c.blocks_h;
        if (nblk * (unsigned long long)c.blocks_w >= (1ull << 32) || nblk >= (1ull << 31)) return HVC_E_TOO_LARGE;
        hvc::CompK &k = L.comp[i];
        k.bw = c.blocks_w;
        k.bh = c.blocks_h;
        k.nblk = (int)nblk;
        k.tile0 = tile;
        k.magic = c.blocks_w == 1 ? 0u : (unsigned)(((1ull << 32) + (unsigned)c.blocks_w - 1) / (unsigned)c.blocks_w);
        k.qtab = c.qtab;
        k.coef_off = c.coef_offset;
        k.plane_off = c.plane_offset;
        k.stride = c.stride;
        tile += (int)((nblk + HVC_TILE - 1) / HVC_TILE);
        size_t ce = c.coef_offset + (size_t)nblk * 64;
        size_t pe = c.plane_offset + ((size_t)c.blocks_h * 8 - 1) * c.stride + (size_t)c.blocks_w * 8;
        if (ce > L.coef_span) L.coef_span = ce;
        if (pe > L.pixel_span) L.pixel_span = pe;
        L.blocks_per_frame += nblk;
    }
    L.tiles_per_frame = tile;
    return HVC_OK;
}

int check_qtabs(const uint16_t *qtabs, int n_qtabs) {
    if (!qtabs || n_qtabs < 1 || n_qtabs > HVC_MAX_QTABS) return HVC_E_INVALID_ARG;
    for (int i = 0; i < n_qtabs * 64; i++)
        if (qtabs[i] == 0) return HVC_E_RANGE;
    return HVC_OK;
}

// The batch pipelines' orchestrating thread waits most of the call (an upload's end, a chunk's kernels).
// hipEventSynchronize spins -- also on an event created with hipEventBlockingSync, on this ROCm (measured: CPU time =
// wall time) -- and on a box whose processes own a fixed share of CPU time (16 CPUs for one GPU here) a spinning thread
// takes its CPU from the workers that are the bound of the pipeline.  So: poll and sleep, 20 us at first, 200 us from
// the tenth poll on (the waits are milliseconds long).  HVC_EVENT_SPIN=1: hipEventSynchronize (A/B).
static hipError_t wait_event(hipEvent_t e) {
    static const bool spin = [] { const char *v = std::getenv("HVC_EVENT_SPIN"); return v && v[0] == '1'; }();
    if (spin) return hipEventSynchronize(e);
    for (int polls = 0;; polls++) {
        const hipError_t r = hipEventQuery(e);
        if (r != hipErrorNotReady) return r;
        std::this_thread::sleep_for(std::chrono::microseconds(polls < 10 ? 20 : 200));
    }
}

// Host buffers, large batches: four parts; while part k + 1 is uploaded (c->stream), part k runs through the kernels
// (c->stream) and is downloaded (a second thread on c->down_stream: copies to and from pageable memory hold
// their caller), so the link carries both directions at once.  up(f0, cnt) / run(k, f0, cnt) enqueue on c->stream,
// down(f0, cnt, stream) on the stream it is given.
template <class Up, class Run, class Down>
static int overlapped_parts(hvc_ctx *c, int n_frames, Up up, Run run, Down down) {
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

} // namespace

extern "C" {

const char *hvc_version(void) { return "hvc_jpeg 0.1 (gfx950)"; }

const char *hvc_strerror(int code) {
    switch (code) {
    case HVC_OK: return "ok";
    case HVC_E_INVALID_ARG: return "invalid argument";
    case HVC_E_NO_DEVICE: return "no usable gfx950 device";
    case HVC_E_HIP: return "HIP runtime error";
    case HVC_E_ALIGNMENT: return "pointer, offset or stride not aligned";
    case HVC_E_RANGE: return "value out of range";
    case HVC_E_OUT_OF_MEMORY: return "out of device memory";
    case HVC_E_TOO_LARGE: return "plane geometry too large";
    case HVC_E_BAD_JPEG: return "malformed or unsupported JPEG stream";
    case HVC_E_UNSUPPORTED_MARKER: return "unsupported marker code";
    case HVC_E_SYSTEM: return "the system refused a host thread";
    case HVC_E_INTERNAL: return "internal error (C++ exception stopped at the boundary)";
    default: return "unknown hvc error";
    }
}

int hvc_create(hvc_ctx **out, int device) try {
    if (!out) return HVC_E_INVALID_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return HVC_E_NO_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return HVC_E_NO_DEVICE;
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return HVC_E_NO_DEVICE; // gfx950 code objects only
    hvc_ctx *c = new (std::nothrow) hvc_ctx();
    if (!c) return HVC_E_OUT_OF_MEMORY;
    c->device = device;
    DeviceGuard g(device);
    bool ok = g.ok && hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) == hipSuccess &&
              hipEventCreate(&c->ev0) == hipSuccess && hipEventCreate(&c->ev1) == hipSuccess &&
              [&] {
                  for (int i = 0; i < HVC_PROF_RING; i++)
                      if (hipEventCreate(&c->k0[i]) != hipSuccess || hipEventCreate(&c->k1[i]) != hipSuccess) return false;
                  return true;
              }() &&
              hipMalloc((void **)&c->d_fix_count, HVC_FIX_WORDS * sizeof(unsigned)) == hipSuccess &&
              hipMemset(c->d_fix_count, 0, HVC_FIX_WORDS * sizeof(unsigned)) == hipSuccess;
    if (!ok) {
        hvc_destroy(c);
        return HVC_E_NO_DEVICE;
    }
    c->stream = c->own_stream;
    c->have_default_cpus = sched_getaffinity(0, sizeof c->default_cpus, &c->default_cpus) == 0;
    if (const char *env = std::getenv("HVC_HOST_CPUS")) (void)hvc_set_host_cpus(c, env); // an unusable list leaves the threads unrestricted
    *out = c;
    return HVC_OK;
} HVC_ABI_CATCH

void hvc_destroy(hvc_ctx *c) {
    if (!c) return;
    c->pool.shutdown(); // (idle: every call waits for its own tasks)
    DeviceGuard g(c->device);
    // everything this context may still have in flight: the caller's stream (NULL is HIP's default stream -- a
    // stream like any other), its own, and the pipelines' side streams
    (void)hipStreamSynchronize(c->stream);
    if (c->own_stream) (void)hipStreamSynchronize(c->own_stream);
    if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
    if (c->down_stream) (void)hipStreamSynchronize(c->down_stream);
    if (c->side_stream) (void)hipStreamSynchronize(c->side_stream);
    for (int i = 0; i < 3; i++)
        if (c->rd_stream[i]) (void)hipStreamSynchronize(c->rd_stream[i]);
    if (c->d_fix_count) (void)hipFree(c->d_fix_count);
    if (c->d_fix_list) (void)hipFree(c->d_fix_list);
    if (c->d_in) (void)hipFree(c->d_in);
    if (c->d_out) (void)hipFree(c->d_out);
    if (c->d_sums) (void)hipFree(c->d_sums);
    if (c->d_aux) (void)hipFree(c->d_aux);
    if (c->d_aux2) (void)hipFree(c->d_aux2);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    for (int i = 0; i < HVC_PROF_RING; i++) {
        if (c->k0[i]) (void)hipEventDestroy(c->k0[i]);
        if (c->k1[i]) (void)hipEventDestroy(c->k1[i]);
    }
    for (int i = 0; i < hvc_ctx::RING; i++) {
        if (c->h_ring[i]) (void)hipHostFree(c->h_ring[i]);
        if (c->d_ring[i]) (void)hipFree(c->d_ring[i]);
        if (c->d_oring[i]) (void)hipFree(c->d_oring[i]);
        if (c->ev_h2d[i]) (void)hipEventDestroy(c->ev_h2d[i]);
        if (c->ev_kern[i]) (void)hipEventDestroy(c->ev_kern[i]);
        if (c->eh_in[i]) (void)hipHostFree(c->eh_in[i]);
        if (c->eh_out[i]) (void)hipHostFree(c->eh_out[i]);
        if (c->ed_in[i]) (void)hipFree(c->ed_in[i]);
        if (c->ed_out[i]) (void)hipFree(c->ed_out[i]);
        if (c->ev_up[i]) (void)hipEventDestroy(c->ev_up[i]);
        if (c->ev_down[i]) (void)hipEventDestroy(c->ev_down[i]);
        for (int k = 0; k < 3; k++)
            if (c->ev_et[i][k]) (void)hipEventDestroy(c->ev_et[i][k]);
        if (c->ev_gpu[i]) (void)hipEventDestroy(c->ev_gpu[i]);
        if (c->gp_h_ecs[i]) (void)hipHostFree(c->gp_h_ecs[i]);
        if (c->gp_d_ecs[i]) (void)hipFree(c->gp_d_ecs[i]);
        if (c->gp_h_meta[i]) (void)hipHostFree(c->gp_h_meta[i]);
        if (c->gp_d_meta[i]) (void)hipFree(c->gp_d_meta[i]);
        if (c->gp_h_ftabs[i]) (void)hipHostFree(c->gp_h_ftabs[i]);
        if (c->gp_d_ftabs[i]) (void)hipFree(c->gp_d_ftabs[i]);
        if (c->ed_seg[i]) (void)hipFree(c->ed_seg[i]);
        if (c->ed_off[i]) (void)hipFree(c->ed_off[i]);
        if (c->eh_off[i]) (void)hipHostFree(c->eh_off[i]);
    }
    for (int i = 0; i < 4; i++)
        if (c->ev_t[i]) (void)hipEventDestroy(c->ev_t[i]);
    if (c->gd_ecs) (void)hipFree(c->gd_ecs);
    if (c->gd_h_ecs) (void)hipHostFree(c->gd_h_ecs);
    if (c->gd_fcnt) (void)hipFree(c->gd_fcnt);
    if (c->gd_meta) (void)hipFree(c->gd_meta);
    if (c->gd_state) (void)hipFree(c->gd_state);
    if (c->gd_tables) (void)hipFree(c->gd_tables);
    if (c->gd_coefs) (void)hipFree(c->gd_coefs);
    if (c->gd_dcd) (void)hipFree(c->gd_dcd);
    if (c->gd_ftabs) (void)hipFree(c->gd_ftabs);
    if (c->gd_dcv) (void)hipFree(c->gd_dcv);
    if (c->d_dcfix) (void)hipFree(c->d_dcfix);
    delete c->gd_tables_host;
    if (c->hd_tables) (void)hipFree(c->hd_tables);
    if (c->hd_lens) (void)hipFree(c->hd_lens);
    if (c->hd_meta) (void)hipFree(c->hd_meta);
    if (c->hd_bitbuf) (void)hipFree(c->hd_bitbuf);
    if (c->hd_ff) (void)hipFree(c->hd_ff);
    if (c->hd_out) (void)hipFree(c->hd_out);
    for (int i = 0; i < 3; i++)
        if (c->rd_stream[i]) (void)hipStreamDestroy(c->rd_stream[i]);
    for (int i = 0; i < 3; i++)
        if (c->ev_rd[i]) (void)hipEventDestroy(c->ev_rd[i]);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    if (c->down_stream) (void)hipStreamDestroy(c->down_stream);
    if (c->side_stream) (void)hipStreamDestroy(c->side_stream);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

int hvc_last_hip_error(const hvc_ctx *c) { return c ? c->last_hip : 0; }

int hvc_set_host_cpus(hvc_ctx *c, const char *cpulist) try {
    if (!c) return HVC_E_INVALID_ARG;
    if (!cpulist || !*cpulist) {
        c->have_cpus = false;
        c->cpulist[0] = 0;
        return HVC_OK;
    }
    char buf[sizeof c->cpulist];
    if (!std::strcmp(cpulist, "auto")) { // the CPUs of the NUMA node the context's GPU hangs off (sysfs, through its PCI address)
        DeviceGuard g(c->device);
        char bus[64] = "", path[160];
        if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, c->device) != hipSuccess) return HVC_E_INVALID_ARG;
        for (char *p = bus; *p; p++) *p = (char)((*p >= 'A' && *p <= 'Z') ? *p - 'A' + 'a' : *p); // sysfs spells it in lower case
        std::snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/local_cpulist", bus);
        FILE *f = std::fopen(path, "r");
        if (!f) return HVC_E_INVALID_ARG;
        const bool got = std::fgets(buf, (int)sizeof buf, f) != nullptr;
        std::fclose(f);
        if (!got) return HVC_E_INVALID_ARG;
        cpulist = buf;
    }
    cpu_set_t set, allowed;
    if (std::strlen(cpulist) >= sizeof c->cpulist || !parse_cpulist(cpulist, set)) return HVC_E_INVALID_ARG;
    // only CPUs this process may use at all (a container's cpuset): an empty intersection is an error
    if (sched_getaffinity(0, sizeof allowed, &allowed) == 0) {
        CPU_AND(&set, &set, &allowed);
        if (CPU_COUNT(&set) == 0) return HVC_E_INVALID_ARG;
    }
    c->cpus = set;
    c->have_cpus = true;
    std::snprintf(c->cpulist, sizeof c->cpulist, "%s", cpulist);
    for (char *p = c->cpulist; *p; p++)
        if (*p == '\n') *p = 0;
    return HVC_OK;
} HVC_ABI_CATCH

int hvc_host_threads(const hvc_ctx *c, int *alive, uint64_t *ever_started) try {
    if (!c) return HVC_E_INVALID_ARG;
    if (alive) *alive = c->pool.size();
    if (ever_started) *ever_started = c->pool.threads_created();
    return HVC_OK;
} HVC_ABI_CATCH

int hvc_host_threads_probe(int threads) try {
    if (threads < 1 || threads > 4096) return HVC_E_INVALID_ARG;
    hvc::WorkerPool pool;
    int r = pool.ensure(threads);
    if (r) return r; // (the pool's destructor joins whatever did start)
    std::atomic<int> ran{0};
    if ((r = pool.submit([&] { ran++; }, threads))) {
        (void)pool.wait();
        return r;
    }
    r = pool.wait();
    return r ? r : ran.load() == threads ? HVC_OK : HVC_E_INTERNAL;
} HVC_ABI_CATCH

int hvc_get_host_cpus(const hvc_ctx *c, char *out, size_t cap, int *n_cpus) try {
    if (!c || (!out && cap)) return HVC_E_INVALID_ARG;
    if (out && cap) std::snprintf(out, cap, "%s", c->have_cpus ? c->cpulist : "");
    if (n_cpus) *n_cpus = c->have_cpus ? CPU_COUNT(&c->cpus) : 0;
    return HVC_OK;
} HVC_ABI_CATCH

// Work enqueued on the stream the context leaves is drained first: device-memory calls return while their kernels
// run, the scratch they use (fix-up list and counters, staging buffers) is re-grown and re-used in the order of ONE
// stream, and grow() only synchronises the current one.
static int switch_stream(hvc_ctx *c, hipStream_t s) {
    if (s == c->stream) return HVC_OK;
    DeviceGuard g(c->device);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->stream = s;
    return HVC_OK;
}

int hvc_set_stream(hvc_ctx *c, void *s) try {
    if (!c) return HVC_E_INVALID_ARG;
    return switch_stream(c, (hipStream_t)s); // NULL is a stream too: HIP's default (null) stream
} HVC_ABI_CATCH

int hvc_reset_stream(hvc_ctx *c) try {
    if (!c) return HVC_E_INVALID_ARG;
    return switch_stream(c, c->own_stream);
} HVC_ABI_CATCH

int hvc_synchronize(hvc_ctx *c) try {
    if (!c) return HVC_E_INVALID_ARG;
    DeviceGuard g(c->device);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return HVC_OK;
} HVC_ABI_CATCH

int hvc_timer_begin(hvc_ctx *c) try {
    if (!c) return HVC_E_INVALID_ARG;
    DeviceGuard g(c->device);
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    return HVC_OK;
} HVC_ABI_CATCH

int hvc_timer_end(hvc_ctx *c, float *ms) try {
    if (!c || !ms) return HVC_E_INVALID_ARG;
    DeviceGuard g(c->device);
    HIPCHK(c, hipEventRecord(c->ev1, c->stream));
    HIPCHK(c, hipEventSynchronize(c->ev1));
    HIPCHK(c, hipEventElapsedTime(ms, c->ev0, c->ev1));
    return HVC_OK;
} HVC_ABI_CATCH

int hvc_set_decode_kernel(hvc_ctx *c, int which) try {
    if (!c || which < 0 || which > 3) return HVC_E_INVALID_ARG;
    c->decode_kernel = which;
    return HVC_OK;
} HVC_ABI_CATCH

int hvc_set_profiling(hvc_ctx *c, int enabled) try {
    if (!c) return HVC_E_INVALID_ARG;
    c->profiling = enabled != 0;
    c->k_calls = 0;
    return HVC_OK;
} HVC_ABI_CATCH

int hvc_kernel_ms_history(hvc_ctx *c, float *ms, int n) try {
    if (!c || !ms || n < 1 || n > HVC_PROF_RING || (unsigned long long)n > c->k_calls) return HVC_E_INVALID_ARG;
    DeviceGuard g(c->device);
    for (int i = 0; i < n; i++) { // ms[0] = oldest of the last n profiled calls
        int slot = (int)((c->k_calls - (unsigned long long)n + (unsigned long long)i) % HVC_PROF_RING);
        HIPCHK(c, hipEventSynchronize(c->k1[slot]));
        HIPCHK(c, hipEventElapsedTime(&ms[i], c->k0[slot], c->k1[slot]));
    }
    return HVC_OK;
} HVC_ABI_CATCH

int hvc_last_kernel_ms(hvc_ctx *c, float *ms) { return hvc_kernel_ms_history(c, ms, 1); }

int hvc_device_alloc(hvc_ctx *c, size_t bytes, void **out) try {
    if (!c || !out) return HVC_E_INVALID_ARG;
    DeviceGuard g(c->device);
    hipError_t e = hipMalloc(out, bytes ? bytes : 1);
    if (e != hipSuccess) {
        c->last_hip = (int)e;
        *out = nullptr;
        return HVC_E_OUT_OF_MEMORY;
    }
    return HVC_OK;
} HVC_ABI_CATCH

int hvc_device_free(hvc_ctx *c, void *p) try {
    if (!c) return HVC_E_INVALID_ARG;
    DeviceGuard g(c->device);
    HIPCHK(c, hipFree(p));
    return HVC_OK;
} HVC_ABI_CATCH

int hvc_memcpy_h2d(hvc_ctx *c, void *dst, const void *src, size_t bytes) try {
    if (!c || (!dst && bytes) || (!src && bytes)) return HVC_E_INVALID_ARG;
    DeviceGuard g(c->device);
    HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return HVC_OK;
} HVC_ABI_CATCH

int hvc_memcpy_d2h(hvc_ctx *c, void *dst, const void *src, size_t bytes) try {
    if (!c || (!dst && bytes) || (!src && bytes)) return HVC_E_INVALID_ARG;
    DeviceGuard g(c->device);
    HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return HVC_OK;
} HVC_ABI_CATCH

int hvc_checksum_records(hvc_ctx *c, const void *data, size_t record_bytes, size_t record_stride, int n_records,
                         uint64_t *sums, int where) try {
    if (!c || !sums || n_records < 0 || (!data && n_records && record_bytes)) return HVC_E_INVALID_ARG;
    if (where != HVC_MEM_HOST && where != HVC_MEM_DEVICE) return HVC_E_INVALID_ARG;
    if (n_records == 0) return HVC_OK;
    if (n_records > 65535) return HVC_E_TOO_LARGE;
    if (n_records > 1 && record_stride < record_bytes) return HVC_E_INVALID_ARG;
    DeviceGuard g(c->device);
    if (!g.ok) return fail_hip(c, hipErrorInvalidDevice);
    const size_t sum_bytes = (size_t)n_records * sizeof(unsigned long long);
    const uint8_t *d = (const uint8_t *)data;
    int r;
    if (where == HVC_MEM_HOST) {
        const size_t bytes = (size_t)(n_records - 1) * record_stride + record_bytes;
        if ((r = grow(c, &c->d_in, &c->in_cap, bytes ? bytes : 1))) return r;
        if (bytes) HIPCHK(c, hipMemcpyAsync(c->d_in, data, bytes, hipMemcpyHostToDevice, c->stream));
        d = (const uint8_t *)c->d_in;
    }
    if ((r = grow(c, &c->d_sums, &c->sums_cap, sum_bytes))) return r;
    HIPCHK(c, hvc::launch_checksum(d, record_bytes, record_stride, n_records, (unsigned long long *)c->d_sums, c->stream));
    HIPCHK(c, hipMemcpyAsync(sums, c->d_sums, sum_bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return HVC_OK;
} HVC_ABI_CATCH

int hvc_last_wide_blocks(hvc_ctx *c, uint64_t *count) try {
    if (!c || !count) return HVC_E_INVALID_ARG;
    DeviceGuard g(c->device);
    unsigned long long v = 0; // the total over ALL launches of the last call (a large batch is cut into several)
    HIPCHK(c, hipMemcpyAsync(&v, c->d_fix_count + 2, sizeof v, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    *count = c->wide_host >= 0 ? (uint64_t)c->wide_host : (uint64_t)v;
    return HVC_OK;
} HVC_ABI_CATCH

// Kernel-side forms of the quantiser tables: plain ints, the energy thresholds of the two int32
// kernels, and the packed kernel's per-row operand pairs.
static void prepare_tables(const uint16_t *qtabs, int n_qtabs, int *qt, int *ethr, int *ethr_packed, unsigned *qpair) {
    for (int i = 0; i < n_qtabs * 64; i++) qt[i] = (int)qtabs[i];
    for (int t = 0; t < n_qtabs; t++) {
        // every |coef * q| <= qmax * sqrt(E): the fast kernel accepts E up to (HVC_GUARD_D / qmax)^2
        unsigned qmax = 1;
        for (int i = 0; i < 64; i++) qmax = qtabs[t * 64 + i] > qmax ? qtabs[t * 64 + i] : qmax;
        unsigned long long m = HVC_GUARD_D / qmax;
        unsigned long long thr = m * m;
        ethr[t] = thr > 0x7ffffffeull ? 0x7ffffffe : (int)thr;
        m = HVC_GUARD_D_PACKED / qmax;
        thr = m * m;
        ethr_packed[t] = thr > 0x7ffffffeull ? 0x7ffffffe : (int)thr;
        static const int PAIRS[4][2] = {{HVC_PAIR_A_LO, HVC_PAIR_A_HI}, {HVC_PAIR_B_LO, HVC_PAIR_B_HI},
                                        {HVC_PAIR_C_LO, HVC_PAIR_C_HI}, {HVC_PAIR_Z_LO, HVC_PAIR_Z_HI}}; // hvc_idct_spec.h
        for (int r = 0; r < 8; r++)
            for (int k = 0; k < 4; k++) {
                unsigned lo = qtabs[t * 64 + hvc::HVC_ZF[8 * r + PAIRS[k][0]]];
                unsigned hi = qtabs[t * 64 + hvc::HVC_ZF[8 * r + PAIRS[k][1]]];
                qpair[t * 32 + r * 4 + k] = (lo & 0xffffu) | (hi << 16);
            }
    }
}

// ---------------------------------------------------------------------------
// A block of a batch whose true DC does not fit the int16 record (hvc::WideDc of frame `frame` of the batch): after
// the batch's launches it is recomputed in int64 with that DC -- what the model's 63-bit arithmetic gives.
struct WideFix {
    int frame;
    uint32_t block;
    long long dc;
};

// ids (fix-list encoding of the launch geometry) + DCs -> device scratch; returns pointers into it
static int upload_dcfix(hvc_ctx *c, const std::vector<unsigned> &ids, const std::vector<long long> &dcs, const unsigned **d_count,
                        const unsigned **d_ids, const long long **d_dcs) {
    const size_t n = ids.size();
    const size_t off_ids = 16, off_dcs = (off_ids + n * sizeof(unsigned) + 15) & ~(size_t)15;
    int r = grow(c, &c->d_dcfix, &c->dcfix_cap, off_dcs + n * sizeof(long long));
    if (r) return r;
    const unsigned cnt = (unsigned)n;
    char *base = (char *)c->d_dcfix;
    HIPCHK(c, hipMemcpyAsync(base, &cnt, sizeof cnt, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(base + off_ids, ids.data(), n * sizeof(unsigned), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(base + off_dcs, dcs.data(), n * sizeof(long long), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream)); // (the sources are the caller's vectors; a rare path)
    *d_count = (const unsigned *)base;
    *d_ids = (const unsigned *)(base + off_ids);
    *d_dcs = (const long long *)(base + off_dcs);
    return HVC_OK;
}

static int apply_wide_dc(hvc_ctx *c, const hvc::DecodeParams &P, const std::vector<WideFix> &wide) {
    std::vector<unsigned> ids;
    std::vector<long long> dcs;
    for (const WideFix &w : wide) {
        for (int i = 0; i < P.n_comp; i++) {
            const hvc::CompK &K = P.comp[i];
            const size_t b0 = K.coef_off / 64;
            if (w.block < b0 || w.block >= b0 + (size_t)K.nblk) continue;
            const unsigned b = (unsigned)(w.block - b0);
            ids.push_back(((unsigned)w.frame * (unsigned)P.tiles_per_frame + (unsigned)K.tile0 + b / HVC_TILE) * HVC_TILE + b % HVC_TILE);
            dcs.push_back(w.dc);
            break;
        }
    }
    if (ids.empty()) return HVC_OK;
    const unsigned *d_count, *d_ids;
    const long long *d_dcs;
    int r = upload_dcfix(c, ids, dcs, &d_count, &d_ids, &d_dcs);
    if (r) return r;
    hvc::DecodeParams Q = P;
    Q.dc_plane = nullptr; // (the list carries the DC)
    HIPCHK(c, hvc::launch_decode_dcfix(Q, d_count, d_ids, d_dcs, c->stream));
    return HVC_OK;
}

static int apply_wide_dc_444(hvc_ctx *c, const hvc::Decode444Params &P, const std::vector<WideFix> &wide) {
    std::vector<unsigned> ids;
    std::vector<long long> dcs;
    const unsigned wgs = (unsigned)(HVC_TILE * P.nw), tw = (unsigned)(HVC_444_TILE_BW * P.nw);
    for (const WideFix &w : wide) {
        int p = -1; // the plane whose coefficient offset is the largest one not beyond the block
        for (int i = 0; i < 3; i++)
            if (P.pl[i].coef_off / 64 <= w.block && (p < 0 || P.pl[i].coef_off > P.pl[p].coef_off)) p = i;
        if (p < 0) continue;
        {
            const hvc::Plane444K &K = P.pl[p];
            const size_t rel = w.block - K.coef_off / 64;
            const unsigned by = (unsigned)(rel / (unsigned)K.bw), bx = (unsigned)(rel % (unsigned)K.bw);
            if (bx >= (unsigned)K.cbw || by >= (unsigned)K.cbh) continue; // outside the crop: never decoded
            unsigned tile, lane;
            if (p == 0) {
                const unsigned b = by * (unsigned)K.cbw + bx;
                tile = b / wgs;
                lane = b % wgs;
            } else { // locate444's chroma mapping, inverted
                unsigned tx = P.c_tiles_x == 1 ? 0u : bx / (tw - 1);
                if (tx >= (unsigned)P.c_tiles_x) tx = (unsigned)P.c_tiles_x - 1;
                const unsigned lx = bx - tx * (tw - 1), ty = by / HVC_444_TILE_BH, ly = by % HVC_444_TILE_BH;
                tile = (unsigned)P.y_tiles + (unsigned)(p - 1) * (unsigned)(P.c_tiles_x * P.c_tiles_y) + ty * (unsigned)P.c_tiles_x + tx;
                lane = ly * tw + lx;
            }
            ids.push_back(((unsigned)w.frame * (unsigned)P.tiles_per_frame + tile) * wgs + lane);
            dcs.push_back(w.dc);
        }
    }
    if (ids.empty()) return HVC_OK;
    const unsigned *d_count, *d_ids;
    const long long *d_dcs;
    int r = upload_dcfix(c, ids, dcs, &d_count, &d_ids, &d_dcs);
    if (r) return r;
    hvc::Decode444Params Q = P;
    Q.dc_plane = nullptr;
    HIPCHK(c, hvc::launch_decode_444_dcfix(Q, d_count, d_ids, d_dcs, (unsigned)ids.size(), c->stream));
    return HVC_OK;
}

// dc_plane (device memory calls only, default kernels only): see hvc::DecodeParams::dc_plane
// wide (device memory calls only): blocks to recompute with their true DC once the launches are enqueued
static int decode_frames_impl(hvc_ctx *c, const int16_t *coefs, size_t coef_fs, const uint16_t *qtabs, int n_qtabs,
                              const hvc_component *comps, int n_comp, int n_frames, uint8_t *pixels, size_t pixel_fs,
                              int where, const int16_t *dc_plane, size_t dc_fs, const std::vector<WideFix> *wide = nullptr) {
    if (!c || !coefs || !pixels || n_frames < 0) return HVC_E_INVALID_ARG;
    if (where != HVC_MEM_HOST && where != HVC_MEM_DEVICE) return HVC_E_INVALID_ARG;
    int r = check_qtabs(qtabs, n_qtabs);
    if (r) return r;
    Layout L;
    r = make_layout(comps, n_comp, n_qtabs, L);
    if (r) return r;
    if (n_frames == 0) return HVC_OK;
    if (n_frames > 65535) return HVC_E_TOO_LARGE;
    if (n_frames > 1 && (coef_fs < L.coef_span || pixel_fs < L.pixel_span)) return HVC_E_INVALID_ARG;
    if ((coef_fs & 7) || (pixel_fs & 7)) return HVC_E_ALIGNMENT;
    unsigned long long ids = (unsigned long long)n_frames * L.tiles_per_frame * HVC_TILE;
    if (ids >= (1ull << 32)) return HVC_E_TOO_LARGE;
    DeviceGuard g(c->device);
    if (!g.ok) return fail_hip(c, hipErrorInvalidDevice);

    // fix-up list: one entry per block at most
    size_t need = (size_t)ids;
    if (need > c->fix_cap) {
        void *p = c->d_fix_list;
        size_t cap = c->fix_cap * sizeof(unsigned);
        r = grow(c, &p, &cap, need * sizeof(unsigned));
        c->d_fix_list = (unsigned *)p;
        c->fix_cap = cap / sizeof(unsigned);
        if (r) return r;
    }

    hvc::DecodeParams P;
    std::memset(&P, 0, sizeof P);
    P.coef_fs = coef_fs;
    P.pixel_fs = pixel_fs;
    P.n_frames = n_frames;
    P.n_comp = L.n_comp;
    P.tiles_per_frame = L.tiles_per_frame;
    for (int i = 0; i < L.n_comp; i++) P.comp[i] = L.comp[i];
    prepare_tables(qtabs, n_qtabs, P.qt, P.ethr, P.ethr_packed, P.qpair);
    wide_total_begin(c);
    fix_assign(c, P);

    // 16-bit quantiser entries above 255 leave the fast kernel's proven range
    // (|coef * q| must stay below 2^23): such planes go straight to the wide kernel.
    bool wide_only = false;
    for (int i = 0; i < n_qtabs * 64; i++) wide_only |= qtabs[i] > 255;
    wide_only |= c->decode_kernel == 2;
    P.kernel_sel = (c->decode_kernel == 1 || c->decode_kernel == 3) ? c->decode_kernel : 0;
    // one launch: consumes counter fix_phase, its wide kernel clears the other one (fix_assign / fix_commit above)
    auto launch = [&](hvc::DecodeParams &Q, hipEvent_t k0, hipEvent_t k1) -> hipError_t {
        if (wide_only) {
            c->wide_host = (long long)((unsigned long long)n_frames * L.blocks_per_frame);
            return hvc::launch_decode_wide_only(Q, c->stream);
        }
        fix_assign(c, Q);
        const hipError_t e = hvc::launch_decode(Q, c->stream, k0, k1);
        if (e == hipSuccess) fix_commit(c);
        else fix_reset(c);
        return e;
    };

    if (where == HVC_MEM_DEVICE) {
        if (((uintptr_t)coefs & 15) || ((uintptr_t)pixels & 7)) return HVC_E_ALIGNMENT;
        if (dc_plane && P.kernel_sel != 0) return HVC_E_INVALID_ARG;
        P.coefs = coefs;
        P.pixels = pixels;
        P.dc_plane = dc_plane;
        P.dc_fs = dc_fs;
        const bool prof = c->profiling && !wide_only;
        const int slot = (int)(c->k_calls % HVC_PROF_RING);
        const int per = frames_per_launch(n_frames, L.blocks_per_frame); // (see launch_bytes_limit)
        for (int f0 = 0; f0 < n_frames; f0 += per) {
            hvc::DecodeParams Pk = P;
            Pk.n_frames = n_frames - f0 < per ? n_frames - f0 : per;
            Pk.coefs = coefs + (size_t)f0 * coef_fs;
            Pk.pixels = pixels + (size_t)f0 * pixel_fs;
            if (dc_plane) Pk.dc_plane = dc_plane + (size_t)f0 * dc_fs;
            // the event pair brackets the dominant kernel of all parts (and the few-microsecond fix-up kernels in between)
            HIPCHK(c, launch(Pk, prof && f0 == 0 ? c->k0[slot] : nullptr, prof && f0 + per >= n_frames ? c->k1[slot] : nullptr));
        }
        if (prof) c->k_calls++;
        if (wide && !wide->empty()) return apply_wide_dc(c, P, *wide);
        return HVC_OK;
    }
    if (dc_plane || (wide && !wide->empty())) return HVC_E_INVALID_ARG;

    // host memory: mirror the caller's record layout on the device
    size_t cbytes = ((size_t)(n_frames - 1) * coef_fs + L.coef_span) * sizeof(int16_t);
    size_t pbytes = (size_t)(n_frames - 1) * pixel_fs + L.pixel_span;
    r = grow(c, &c->d_in, &c->in_cap, cbytes);
    if (r) return r;
    r = grow(c, &c->d_out, &c->out_cap, pbytes);
    if (r) return r;
    // copy back only the pixels the kernels wrote (the caller's padding stays untouched).  Tight planes that
    // follow one another -- the usual record -- are one stretch per frame, or one for the whole batch.
    size_t run0 = 0, run = 0; // [run0, run0 + run): the planes as one byte range of the record, if they are one
    {
        bool tight = true;
        int order[4] = {0, 1, 2, 3};
        for (int i = 0; i < n_comp; i++)
            for (int j = i + 1; j < n_comp; j++)
                if (comps[order[j]].plane_offset < comps[order[i]].plane_offset) std::swap(order[i], order[j]);
        size_t at = comps[order[0]].plane_offset;
        run0 = at;
        for (int i = 0; i < n_comp && tight; i++) {
            const hvc_component &k = comps[order[i]];
            tight = k.plane_offset == at && k.stride == (size_t)k.blocks_w * 8;
            at += (size_t)k.blocks_w * 8 * (size_t)k.blocks_h * 8;
        }
        run = tight ? at - run0 : 0;
    }
    if (run && n_frames >= 8 && cbytes >= ((size_t)64 << 20)) // large batches in that usual form: see overlapped_parts
        return overlapped_parts(
            c, n_frames,
            [&](int f0, int cnt) {
                return hipMemcpyAsync((int16_t *)c->d_in + (size_t)f0 * coef_fs, coefs + (size_t)f0 * coef_fs,
                                      ((size_t)(cnt - 1) * coef_fs + L.coef_span) * sizeof(int16_t), hipMemcpyHostToDevice, c->stream);
            },
            [&](int, int f0, int cnt) {
                hvc::DecodeParams Pk = P;
                Pk.coefs = (const int16_t *)c->d_in + (size_t)f0 * coef_fs;
                Pk.pixels = (uint8_t *)c->d_out + (size_t)f0 * pixel_fs;
                Pk.n_frames = cnt;
                return launch(Pk, nullptr, nullptr);
            },
            [&](int f0, int cnt, hipStream_t st) {
                const size_t off = (size_t)f0 * pixel_fs + run0;
                return pixel_fs == run ? hipMemcpyAsync(pixels + off, (uint8_t *)c->d_out + off, (size_t)cnt * run,
                                                        hipMemcpyDeviceToHost, st)
                                       : hipMemcpy2DAsync(pixels + off, pixel_fs, (uint8_t *)c->d_out + off, pixel_fs, run, (size_t)cnt,
                                                          hipMemcpyDeviceToHost, st);
            });
    HIPCHK(c, hipMemcpyAsync(c->d_in, coefs, cbytes, hipMemcpyHostToDevice, c->stream));
    P.coefs = (const int16_t *)c->d_in;
    P.pixels = (uint8_t *)c->d_out;
    HIPCHK(c, launch(P, nullptr, nullptr));
    if (run && (n_frames == 1 || pixel_fs == run)) {
        HIPCHK(c, hipMemcpyAsync(pixels + run0, (uint8_t *)c->d_out + run0, (size_t)(n_frames - 1) * pixel_fs + run,
                                 hipMemcpyDeviceToHost, c->stream));
    } else if (run) {
        HIPCHK(c, hipMemcpy2DAsync(pixels + run0, pixel_fs, (uint8_t *)c->d_out + run0, pixel_fs, run, (size_t)n_frames,
                                   hipMemcpyDeviceToHost, c->stream));
    } else {
        for (int f = 0; f < n_frames; f++)
            for (int i = 0; i < n_comp; i++) {
                size_t off = (size_t)f * pixel_fs + comps[i].plane_offset;
                HIPCHK(c, hipMemcpy2DAsync(pixels + off, comps[i].stride, (uint8_t *)c->d_out + off, comps[i].stride,
                                           (size_t)comps[i].blocks_w * 8, (size_t)comps[i].blocks_h * 8,
                                           hipMemcpyDeviceToHost, c->stream));
            }
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return HVC_OK;
}

int hvc_decode_frames(hvc_ctx *c, const int16_t *coefs, size_t coef_fs, const uint16_t *qtabs, int n_qtabs,
                      const hvc_component *comps, int n_comp, int n_frames, uint8_t *pixels, size_t pixel_fs,
                      int where) try {
    return decode_frames_impl(c, coefs, coef_fs, qtabs, n_qtabs, comps, n_comp, n_frames, pixels, pixel_fs, where, nullptr, 0);
} HVC_ABI_CATCH

// ---------------------------------------------------------------------------
// 4:2:0 coefficient records -> tight 4:4:4 frames (block stage + crop + chroma upsample fused)
static int decode_frames_yuv444_impl(hvc_ctx *c, const int16_t *coefs, size_t coef_fs, const uint16_t *qtabs, int n_qtabs,
                                     const hvc_component *comps, int n_comp, int n_frames, int width, int height,
                                     uint8_t *frames, size_t frame_stride, int where, const int16_t *dc_plane, size_t dc_fs,
                                     const std::vector<WideFix> *wide = nullptr) {
    if (!c || !coefs || !frames || !comps || n_frames < 0) return HVC_E_INVALID_ARG;
    if (where != HVC_MEM_HOST && where != HVC_MEM_DEVICE) return HVC_E_INVALID_ARG;
    int r = check_qtabs(qtabs, n_qtabs);
    if (r) return r;
    // Yuv.assert_is_420 (tools/src/yuv.ml:104-116): wy = 2 wu, hy = 2 hu -- even luma size only
    if (n_comp != 3 || width < 2 || height < 2 || (width & 1) || (height & 1)) return HVC_E_INVALID_ARG;
    if (width > 65535 || height > 65535) return HVC_E_TOO_LARGE;
    Layout L; // validates blocks_w / blocks_h / qtab / coef_offset exactly as hvc_decode_frames does
    hvc_component geo[3];
    for (int i = 0; i < 3; i++) { // plane_offset / stride are not used by this entry point: anything goes
        geo[i] = comps[i];
        geo[i].plane_offset = 0;
        geo[i].stride = comps[i].blocks_w > 0 ? (size_t)comps[i].blocks_w * 8 : 0;
    }
    r = make_layout(geo, n_comp, n_qtabs, L);
    if (r) return r;
    const int aw[3] = {width, width / 2, width / 2}, ah[3] = {height, height / 2, height / 2};
    for (int i = 0; i < 3; i++) // the crop must lie inside the decoded planes (decoder.ml:403-413)
        if (comps[i].blocks_w * 8 < aw[i] || comps[i].blocks_h * 8 < ah[i]) return HVC_E_INVALID_ARG;
    const size_t plane_bytes = (size_t)width * (size_t)height, out_span = 3 * plane_bytes;
    if (n_frames == 0) return HVC_OK;
    if (n_frames > 65535) return HVC_E_TOO_LARGE;
    if (n_frames > 1 && (coef_fs < L.coef_span || frame_stride < out_span)) return HVC_E_INVALID_ARG;
    if (coef_fs & 7) return HVC_E_ALIGNMENT;

    hvc::Decode444Params P;
    std::memset(&P, 0, sizeof P);
    P.coef_fs = coef_fs;
    P.out_fs = frame_stride;
    P.n_frames = n_frames;
    P.width = width;
    P.height = height;
    for (int i = 0; i < 3; i++) {
        hvc::Plane444K &K = P.pl[i];
        K.bw = comps[i].blocks_w;
        K.cbw = (aw[i] + 7) / 8;
        K.cbh = (ah[i] + 7) / 8;
        K.aw = aw[i];
        K.ah = ah[i];
        K.qtab = comps[i].qtab;
        K.coef_off = comps[i].coef_offset;
        K.out_off = (size_t)i * plane_bytes;
    }
    // umulhi(b, ceil(2^32 / d)) == b / d needs b * d < 2^32
    if ((unsigned long long)P.pl[0].cbw * P.pl[0].cbw * P.pl[0].cbh >= (1ull << 32)) return HVC_E_TOO_LARGE;
    // the 16-byte store form (and with it the wide chroma tiles) needs aligned rows: device output as the caller gave it,
    // host output through the library's own (256-byte aligned) scratch
    const bool aligned = width % 16 == 0 && frame_stride % 16 == 0 && (where == HVC_MEM_HOST || (uintptr_t)frames % 16 == 0);
    hvc::plan_decode_444(P, aligned);
    const unsigned long long ids = (unsigned long long)n_frames * P.tiles_per_frame * HVC_TILE * P.nw;
    if (ids >= (1ull << 32)) return HVC_E_TOO_LARGE;
    DeviceGuard g(c->device);
    if (!g.ok) return fail_hip(c, hipErrorInvalidDevice);
    if ((size_t)ids > c->fix_cap) {
        void *p = c->d_fix_list;
        size_t cap = c->fix_cap * sizeof(unsigned);
        r = grow(c, &p, &cap, (size_t)ids * sizeof(unsigned));
        c->d_fix_list = (unsigned *)p;
        c->fix_cap = cap / sizeof(unsigned);
        if (r) return r;
    }
    int ethr_unused[HVC_MAX_QTABS];
    prepare_tables(qtabs, n_qtabs, P.qt, ethr_unused, P.ethr_packed, P.qpair);
    bool wide_only = c->decode_kernel == 2;
    for (int i = 0; i < n_qtabs * 64; i++) wide_only |= qtabs[i] > 255;
    P.fix_list = c->d_fix_list;
    wide_total_begin(c);
    // How the block stage is launched (fused444_mode): one kernel for luma and chroma tiles, or -- where the luma crop is
    // whole blocks of the whole coefficient plane, which is every frame whose width and height are multiples of 16 and
    // 8 -- the luma planes through k_decode_packed ITSELF (one component, stride = width: the kernel, the schedule and
    // the fix-up kernel of hvc_decode_frames) and the chroma tiles alone in k_decode_444, one after the other or side
    // by side on two streams.
    const int split = (aligned && !wide_only && c->decode_kernel == 0 && height % 8 == 0 && P.pl[0].cbw == P.pl[0].bw) ? fused444_mode() : 0;
    const size_t luma_ids = split ? (size_t)n_frames * (size_t)((P.pl[0].cbw * P.pl[0].cbh + HVC_TILE - 1) / HVC_TILE) * HVC_TILE : 0;
    if (luma_ids + (size_t)ids > c->fix_cap) { // (the luma list sits behind the 4:4:4 kernels' list)
        void *p = c->d_fix_list;
        size_t cap = c->fix_cap * sizeof(unsigned);
        r = grow(c, &p, &cap, (luma_ids + (size_t)ids) * sizeof(unsigned));
        c->d_fix_list = (unsigned *)p;
        c->fix_cap = cap / sizeof(unsigned);
        if (r) return r;
        P.fix_list = c->d_fix_list;
    }
    if (split == 2 && !c->side_stream) {
        // (streams of one priority may share a hardware queue and then never overlap: HVC_444_SIDE_PRIO -1 / 0 / 1 =
        // highest / the default / lowest priority for the side stream, experiments)
        static const int prio_sel = [] { const char *v = std::getenv("HVC_444_SIDE_PRIO"); return v ? std::atoi(v) : 0; }();
        int least = 0, greatest = 0;
        HIPCHK(c, hipDeviceGetStreamPriorityRange(&least, &greatest));
        HIPCHK(c, hipStreamCreateWithPriority(&c->side_stream, hipStreamNonBlocking, prio_sel < 0 ? greatest : prio_sel > 0 ? least : 0));
        HIPCHK(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
        HIPCHK(c, hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
    }
    auto launch_split = [&](hvc::Decode444Params &Q, hipEvent_t k0, hipEvent_t k1) -> hipError_t {
        const hvc::Plane444K &K = Q.pl[0];
        hvc::DecodeParams Y;
        std::memset(&Y, 0, sizeof Y);
        Y.coefs = Q.coefs;
        Y.pixels = Q.out;
        Y.coef_fs = Q.coef_fs;
        Y.pixel_fs = Q.out_fs;
        Y.n_frames = Q.n_frames;
        Y.n_comp = 1;
        Y.comp[0].bw = K.cbw;
        Y.comp[0].bh = K.cbh;
        Y.comp[0].nblk = K.cbw * K.cbh;
        Y.comp[0].magic = K.cbw == 1 ? 0u : (unsigned)(((1ull << 32) + (unsigned)K.cbw - 1) / (unsigned)K.cbw);
        Y.comp[0].qtab = K.qtab;
        Y.comp[0].coef_off = K.coef_off;
        Y.comp[0].plane_off = K.out_off;
        Y.comp[0].stride = (size_t)Q.width;
        Y.tiles_per_frame = (Y.comp[0].nblk + HVC_TILE - 1) / HVC_TILE;
        prepare_tables(qtabs, n_qtabs, Y.qt, Y.ethr, Y.ethr_packed, Y.qpair);
        Y.dc_plane = Q.dc_plane;
        Y.dc_fs = Q.dc_fs;
        Y.fix_count = c->d_fix_count + 4 + c->fix_phase_l;
        Y.fix_count_next = c->d_fix_count + 4 + (c->fix_phase_l ^ 1);
        Y.fix_list = c->d_fix_list + (size_t)ids;
        Y.wide_total = reinterpret_cast<unsigned long long *>(c->d_fix_count + 2);
        fix_assign(c, Q);
        Q.tile0 = Q.y_tiles; // the chroma tiles alone
        hipStream_t ys = c->stream;
        hipError_t e = hipSuccess;
        if (k0) e = hipEventRecord(k0, c->stream);
        if (split == 2) { // side by side: the total is cleared once per call, both fix-up kernels add atomically
            if (e == hipSuccess && !c->wide_total_started) e = hipMemsetAsync(Y.wide_total, 0, sizeof(unsigned long long), c->stream);
            Y.wide_first = Q.wide_first = 2;
            if (e == hipSuccess) e = hipEventRecord(c->ev_fork, c->stream);
            if (e == hipSuccess) e = hipStreamWaitEvent(c->side_stream, c->ev_fork, 0);
            ys = c->side_stream;
        } else {
            Y.wide_first = Q.wide_first; // luma first: it starts the call's total where this is the call's first launch
            Q.wide_first = 0;
        }
        if (e == hipSuccess) e = hvc::launch_decode(Y, ys, nullptr, nullptr);
        if (e == hipSuccess) c->fix_phase_l ^= 1;
        if (e == hipSuccess) e = hvc::launch_decode_444(Q, false, c->stream, nullptr, nullptr);
        if (e == hipSuccess && split == 2) {
            e = hipEventRecord(c->ev_join, c->side_stream);
            if (e == hipSuccess) e = hipStreamWaitEvent(c->stream, c->ev_join, 0);
        }
        if (e == hipSuccess && k1) e = hipEventRecord(k1, c->stream);
        if (e == hipSuccess) fix_commit(c);
        else {
            if (c->side_stream) (void)hipStreamSynchronize(c->side_stream);
            fix_reset(c);
        }
        return e;
    };
    auto launch = [&](hvc::Decode444Params &Q, hipEvent_t k0, hipEvent_t k1) -> hipError_t {
        if (wide_only) { // no list, no counters: every block inside the crop
            c->wide_host = (long long)n_frames * ((long long)P.pl[0].cbw * P.pl[0].cbh + 2ll * P.pl[1].cbw * P.pl[1].cbh);
            return hvc::launch_decode_444(Q, true, c->stream, k0, k1);
        }
        if (split) return launch_split(Q, k0, k1);
        fix_assign(c, Q);
        const hipError_t e = hvc::launch_decode_444(Q, false, c->stream, k0, k1);
        if (e == hipSuccess) fix_commit(c);
        else fix_reset(c);
        return e;
    };

    if (where == HVC_MEM_HOST && (dc_plane || (wide && !wide->empty()))) return HVC_E_INVALID_ARG;
    if (where == HVC_MEM_DEVICE) {
        if ((uintptr_t)coefs & 15) return HVC_E_ALIGNMENT;
        P.coefs = coefs;
        P.out = frames;
        P.dc_plane = dc_plane;
        P.dc_fs = dc_fs;
        const bool prof = c->profiling && !wide_only;
        const int slot = (int)(c->k_calls % HVC_PROF_RING);
        const int per = frames_per_launch(n_frames, L.blocks_per_frame); // (see launch_bytes_limit)
        for (int f0 = 0; f0 < n_frames; f0 += per) {
            hvc::Decode444Params Pk = P;
            Pk.n_frames = n_frames - f0 < per ? n_frames - f0 : per;
            Pk.coefs = coefs + (size_t)f0 * coef_fs;
            Pk.out = frames + (size_t)f0 * frame_stride;
            if (dc_plane) Pk.dc_plane = dc_plane + (size_t)f0 * dc_fs;
            HIPCHK(c, launch(Pk, prof && f0 == 0 ? c->k0[slot] : nullptr, prof && f0 + per >= n_frames ? c->k1[slot] : nullptr));
        }
        if (prof) c->k_calls++;
        if (wide && !wide->empty()) return apply_wide_dc_444(c, P, *wide);
        return HVC_OK;
    }
    const size_t cbytes = ((size_t)(n_frames - 1) * coef_fs + L.coef_span) * sizeof(int16_t);
    const size_t obytes = (size_t)(n_frames - 1) * frame_stride + out_span;
    r = grow(c, &c->d_in, &c->in_cap, cbytes);
    if (r) return r;
    r = grow(c, &c->d_out, &c->out_cap, obytes);
    if (r) return r;
    if (n_frames >= 8 && cbytes >= ((size_t)64 << 20)) // large batches: see overlapped_parts
        return overlapped_parts(
            c, n_frames,
            [&](int f0, int cnt) {
                return hipMemcpyAsync((int16_t *)c->d_in + (size_t)f0 * coef_fs, coefs + (size_t)f0 * coef_fs,
                                      ((size_t)(cnt - 1) * coef_fs + L.coef_span) * sizeof(int16_t), hipMemcpyHostToDevice, c->stream);
            },
            [&](int, int f0, int cnt) {
                auto Pk = P;
                Pk.coefs = (const int16_t *)c->d_in + (size_t)f0 * coef_fs;
                Pk.out = (uint8_t *)c->d_out + (size_t)f0 * frame_stride;
                Pk.n_frames = cnt;
                return launch(Pk, nullptr, nullptr);
            },
            [&](int f0, int cnt, hipStream_t st) {
                const size_t off = (size_t)f0 * frame_stride;
                return hipMemcpy2DAsync(frames + off, frame_stride, (uint8_t *)c->d_out + off, frame_stride, out_span, (size_t)cnt,
                                        hipMemcpyDeviceToHost, st);
            });
    HIPCHK(c, hipMemcpyAsync(c->d_in, coefs, cbytes, hipMemcpyHostToDevice, c->stream));
    P.coefs = (const int16_t *)c->d_in;
    P.out = (uint8_t *)c->d_out;
    HIPCHK(c, launch(P, nullptr, nullptr));
    if (n_frames == 1) // frame_stride is irrelevant for a single frame (and may be smaller than the frame)
        HIPCHK(c, hipMemcpyAsync(frames, c->d_out, out_span, hipMemcpyDeviceToHost, c->stream));
    else
        HIPCHK(c, hipMemcpy2DAsync(frames, frame_stride, c->d_out, frame_stride, out_span, (size_t)n_frames,
                                   hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return HVC_OK;
}

int hvc_decode_frames_yuv444(hvc_ctx *c, const int16_t *coefs, size_t coef_fs, const uint16_t *qtabs, int n_qtabs,
                             const hvc_component *comps, int n_comp, int n_frames, int width, int height,
                             uint8_t *frames, size_t frame_stride, int where) try {
    return decode_frames_yuv444_impl(c, coefs, coef_fs, qtabs, n_qtabs, comps, n_comp, n_frames, width, height, frames,
                                     frame_stride, where, nullptr, 0);
} HVC_ABI_CATCH

int hvc_dequant_idct_recon(hvc_ctx *c, const int16_t *coefs, size_t coef_plane_stride, const uint16_t *qtab,
                           int blocks_w, int blocks_h, int n_planes, uint8_t *plane, size_t stride,
                           size_t plane_stride, int where) try {
    if (blocks_w < 1 || blocks_h < 1) return HVC_E_INVALID_ARG;
    hvc_component comp;
    std::memset(&comp, 0, sizeof comp);
    comp.blocks_w = blocks_w;
    comp.blocks_h = blocks_h;
    comp.qtab = 0;
    comp.stride = stride;
    if (!coef_plane_stride) coef_plane_stride = (size_t)blocks_w * blocks_h * 64;
    if (!plane_stride) plane_stride = stride * (size_t)blocks_h * 8;
    // planes are "frames" of one component; split batches beyond the grid.y limit
    int done = 0;
    while (done < n_planes) {
        int n = n_planes - done > 65535 ? 65535 : n_planes - done;
        int r = hvc_decode_frames(c, coefs + (size_t)done * coef_plane_stride, coef_plane_stride, qtab, 1, &comp, 1,
                                  n, plane + (size_t)done * plane_stride, plane_stride, where);
        if (r) return r;
        done += n;
    }
    return n_planes < 0 ? HVC_E_INVALID_ARG : HVC_OK;
} HVC_ABI_CATCH

// ---------------------------------------------------------------------------
int hvc_encode_frames(hvc_ctx *c, const uint8_t *pixels, size_t pixel_fs, const uint16_t *qtabs, int n_qtabs,
                      const hvc_component *comps, int n_comp, int n_frames, int16_t *coefs, size_t coef_fs,
                      int where) try {
    if (!c || !coefs || !pixels || n_frames < 0) return HVC_E_INVALID_ARG;
    if (where != HVC_MEM_HOST && where != HVC_MEM_DEVICE) return HVC_E_INVALID_ARG;
    int r = check_qtabs(qtabs, n_qtabs);
    if (r) return r;
    // Encoder tables are 8-bit (Markers.Dqt element_precision = 8, encoder.ml:224-229;
    // Quant_tables.scale clips to 1..255, quant_tables.ml:139-147).
    for (int i = 0; i < n_qtabs * 64; i++)
        if (qtabs[i] > 255) return HVC_E_RANGE;
    Layout L;
    r = make_layout(comps, n_comp, n_qtabs, L);
    if (r) return r;
    if (n_frames == 0) return HVC_OK;
    if (n_frames > 65535) return HVC_E_TOO_LARGE;
    if (n_frames > 1 && (coef_fs < L.coef_span || pixel_fs < L.pixel_span)) return HVC_E_INVALID_ARG;
    if ((coef_fs & 7) || (pixel_fs & 7)) return HVC_E_ALIGNMENT;
    DeviceGuard g(c->device);
    if (!g.ok) return fail_hip(c, hipErrorInvalidDevice);

    hvc::EncodeParams P;
    std::memset(&P, 0, sizeof P);
    P.coef_fs = coef_fs;
    P.pixel_fs = pixel_fs;
    P.n_frames = n_frames;
    P.n_comp = L.n_comp;
    P.tiles_per_frame = L.tiles_per_frame;
    for (int i = 0; i < L.n_comp; i++) P.comp[i] = L.comp[i];
    for (int i = 0; i < n_qtabs * 64; i++) // fl((1 + 2^-16) / (4t)): see quant1 in hvc_kernels.hip
        P.qrcp[i] = (float)((1.0 + 1.0 / 65536.0) / (4.0 * (double)qtabs[i]));

    if (where == HVC_MEM_DEVICE) {
        if (((uintptr_t)coefs & 15) || ((uintptr_t)pixels & 7)) return HVC_E_ALIGNMENT;
        const bool prof = c->profiling;
        const int slot = (int)(c->k_calls % HVC_PROF_RING);
        const int per = frames_per_launch(n_frames, L.blocks_per_frame); // (see launch_bytes_limit)
        for (int f0 = 0; f0 < n_frames; f0 += per) {
            hvc::EncodeParams Pk = P;
            Pk.n_frames = n_frames - f0 < per ? n_frames - f0 : per;
            Pk.coefs = coefs + (size_t)f0 * coef_fs;
            Pk.pixels = pixels + (size_t)f0 * pixel_fs;
            HIPCHK(c, hvc::launch_encode(Pk, c->stream, prof && f0 == 0 ? c->k0[slot] : nullptr,
                                         prof && f0 + per >= n_frames ? c->k1[slot] : nullptr));
        }
        if (prof) c->k_calls++;
        return HVC_OK;
    }

    size_t cbytes = ((size_t)(n_frames - 1) * coef_fs + L.coef_span) * sizeof(int16_t);
    size_t pbytes = (size_t)(n_frames - 1) * pixel_fs + L.pixel_span;
    r = grow(c, &c->d_in, &c->in_cap, pbytes);
    if (r) return r;
    r = grow(c, &c->d_out, &c->out_cap, cbytes);
    if (r) return r;
    // copy back only the coefficient planes (gaps in the caller's records stay untouched); planes that follow one
    // another -- the usual record -- are one stretch per frame
    size_t run0 = 0, run = 0; // in int16 elements
    {
        bool adjacent = true;
        int order[4] = {0, 1, 2, 3};
        for (int i = 0; i < n_comp; i++)
            for (int j = i + 1; j < n_comp; j++)
                if (comps[order[j]].coef_offset < comps[order[i]].coef_offset) std::swap(order[i], order[j]);
        size_t at = comps[order[0]].coef_offset;
        run0 = at;
        for (int i = 0; i < n_comp && adjacent; i++) {
            adjacent = comps[order[i]].coef_offset == at;
            at += (size_t)comps[order[i]].blocks_w * comps[order[i]].blocks_h * 64;
        }
        run = adjacent ? at - run0 : 0;
    }
    if (run && n_frames >= 8 && cbytes >= ((size_t)64 << 20)) // large batches in that usual form: see overlapped_parts
        return overlapped_parts(
            c, n_frames,
            [&](int f0, int cnt) {
                return hipMemcpyAsync((uint8_t *)c->d_in + (size_t)f0 * pixel_fs, pixels + (size_t)f0 * pixel_fs,
                                      (size_t)(cnt - 1) * pixel_fs + L.pixel_span, hipMemcpyHostToDevice, c->stream);
            },
            [&](int, int f0, int cnt) {
                hvc::EncodeParams Pk = P;
                Pk.pixels = (const uint8_t *)c->d_in + (size_t)f0 * pixel_fs;
                Pk.coefs = (int16_t *)c->d_out + (size_t)f0 * coef_fs;
                Pk.n_frames = cnt;
                return hvc::launch_encode(Pk, c->stream);
            },
            [&](int f0, int cnt, hipStream_t st) {
                const size_t off = (size_t)f0 * coef_fs + run0;
                return coef_fs == run ? hipMemcpyAsync(coefs + off, (int16_t *)c->d_out + off, (size_t)cnt * run * sizeof(int16_t),
                                                       hipMemcpyDeviceToHost, st)
                                      : hipMemcpy2DAsync(coefs + off, coef_fs * sizeof(int16_t), (int16_t *)c->d_out + off,
                                                         coef_fs * sizeof(int16_t), run * sizeof(int16_t), (size_t)cnt,
                                                         hipMemcpyDeviceToHost, st);
            });
    HIPCHK(c, hipMemcpyAsync(c->d_in, pixels, pbytes, hipMemcpyHostToDevice, c->stream));
    P.pixels = (const uint8_t *)c->d_in;
    P.coefs = (int16_t *)c->d_out;
    HIPCHK(c, hvc::launch_encode(P, c->stream));
    for (int f = 0; f < n_frames; f++)
        for (int i = 0; i < n_comp; i++) {
            size_t off = (size_t)f * coef_fs + comps[i].coef_offset;
            size_t n = (size_t)comps[i].blocks_w * comps[i].blocks_h * 64;
            HIPCHK(c, hipMemcpyAsync(coefs + off, (int16_t *)c->d_out + off, n * sizeof(int16_t),
                                     hipMemcpyDeviceToHost, c->stream));
        }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return HVC_OK;
} HVC_ABI_CATCH

// Encoder.encode_block with compute_reconstruction_error (encoder.ml:195-205): K3, then K1 on the coefficients it
// wrote, then the error plane.  A debugging path in the model and here: three launches, nothing fused.
int hvc_encode_frames_recon(hvc_ctx *c, const uint8_t *pixels, size_t pixel_fs, const uint16_t *qtabs, int n_qtabs,
                            const hvc_component *comps, int n_comp, int n_frames, int16_t *coefs, size_t coef_fs,
                            uint8_t *recon, uint8_t *error, int where) try {
    if (!c || !coefs || !pixels || n_frames < 0) return HVC_E_INVALID_ARG;
    if (where != HVC_MEM_HOST && where != HVC_MEM_DEVICE) return HVC_E_INVALID_ARG;
    Layout L;
    int r = check_qtabs(qtabs, n_qtabs);
    if (!r) r = make_layout(comps, n_comp, n_qtabs, L);
    if (r) return r;
    if (n_frames == 0) return HVC_OK;
    if (n_frames > 1 && (coef_fs < L.coef_span || pixel_fs < L.pixel_span)) return HVC_E_INVALID_ARG;
    DeviceGuard g(c->device);
    if (!g.ok) return fail_hip(c, hipErrorInvalidDevice);
    const size_t pbytes = (size_t)(n_frames - 1) * pixel_fs + L.pixel_span;
    const size_t cbytes = ((size_t)(n_frames - 1) * coef_fs + L.coef_span) * sizeof(int16_t);
    const bool prof_saved = c->profiling;
    struct Restore {
        hvc_ctx *c;
        bool p;
        ~Restore() { c->profiling = p; }
    } restore{c, prof_saved};
    c->profiling = false;
    // device pointers of the three pixel-layout records and of the coefficients
    const uint8_t *d_pix = pixels;
    int16_t *d_coefs = coefs;
    uint8_t *d_recon = recon, *d_error = error;
    if (where == HVC_MEM_HOST) { // [pixels | recon | error] in one scratch allocation, coefficients in another
        const size_t slot = (pbytes + 255) & ~(size_t)255;
        if ((r = grow(c, &c->d_aux, &c->aux_cap, 3 * slot))) return r;
        if ((r = grow(c, &c->d_aux2, &c->aux2_cap, cbytes))) return r;
        d_pix = (const uint8_t *)c->d_aux;
        d_recon = (uint8_t *)c->d_aux + slot;
        d_error = (uint8_t *)c->d_aux + 2 * slot;
        d_coefs = (int16_t *)c->d_aux2;
        HIPCHK(c, hipMemcpyAsync(c->d_aux, pixels, pbytes, hipMemcpyHostToDevice, c->stream));
    } else if (!recon) { // the error plane needs the reconstruction somewhere
        if ((r = grow(c, &c->d_aux, &c->aux_cap, pbytes))) return r;
        d_recon = (uint8_t *)c->d_aux;
    }
    if ((r = hvc_encode_frames(c, d_pix, pixel_fs, qtabs, n_qtabs, comps, n_comp, n_frames, d_coefs, coef_fs, HVC_MEM_DEVICE)))
        return r;
    if ((r = hvc_decode_frames(c, d_coefs, coef_fs, qtabs, n_qtabs, comps, n_comp, n_frames, d_recon, pixel_fs, HVC_MEM_DEVICE)))
        return r;
    if (error || where == HVC_MEM_HOST) {
        hvc::EncodeParams P;
        std::memset(&P, 0, sizeof P);
        P.pixels = d_pix;
        P.pixel_fs = pixel_fs;
        P.n_frames = n_frames;
        P.n_comp = L.n_comp;
        P.tiles_per_frame = L.tiles_per_frame;
        for (int i = 0; i < L.n_comp; i++) P.comp[i] = L.comp[i];
        HIPCHK(c, hvc::launch_abs_error(P, d_recon, d_error, c->stream));
    }
    if (where == HVC_MEM_DEVICE) return HVC_OK;
    for (int f = 0; f < n_frames; f++) // back to the caller: the component planes only (padding stays as it was)
        for (int i = 0; i < n_comp; i++) {
            const size_t coff = (size_t)f * coef_fs + comps[i].coef_offset, cn = (size_t)comps[i].blocks_w * comps[i].blocks_h * 64;
            HIPCHK(c, hipMemcpyAsync(coefs + coff, d_coefs + coff, cn * sizeof(int16_t), hipMemcpyDeviceToHost, c->stream));
            const size_t poff = (size_t)f * pixel_fs + comps[i].plane_offset;
            uint8_t *const dst[2] = {recon, error};
            const uint8_t *const src[2] = {d_recon, d_error};
            for (int k = 0; k < 2; k++)
                if (dst[k])
                    HIPCHK(c, hipMemcpy2DAsync(dst[k] + poff, comps[i].stride, src[k] + poff, comps[i].stride,
                                               (size_t)comps[i].blocks_w * 8, (size_t)comps[i].blocks_h * 8,
                                               hipMemcpyDeviceToHost, c->stream));
        }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return HVC_OK;
} HVC_ABI_CATCH

int hvc_fdct_quant(hvc_ctx *c, const uint8_t *plane, size_t stride, size_t plane_stride, const uint16_t *qtab,
                   int blocks_w, int blocks_h, int n_planes, int16_t *coefs, size_t coef_plane_stride, int where) try {
    if (blocks_w < 1 || blocks_h < 1 || n_planes < 0) return HVC_E_INVALID_ARG;
    hvc_component comp;
    std::memset(&comp, 0, sizeof comp);
    comp.blocks_w = blocks_w;
    comp.blocks_h = blocks_h;
    comp.qtab = 0;
    comp.stride = stride;
    if (!coef_plane_stride) coef_plane_stride = (size_t)blocks_w * blocks_h * 64;
    if (!plane_stride) plane_stride = stride * (size_t)blocks_h * 8;
    int done = 0;
    while (done < n_planes) {
        int n = n_planes - done > 65535 ? 65535 : n_planes - done;
        int r = hvc_encode_frames(c, plane + (size_t)done * plane_stride, plane_stride, qtab, 1, &comp, 1, n,
                                  coefs + (size_t)done * coef_plane_stride, coef_plane_stride, where);
        if (r) return r;
        done += n;
    }
    return HVC_OK;
} HVC_ABI_CATCH

int hvc_upsample420(hvc_ctx *c, const uint8_t *src, int cw, int ch, size_t src_stride, uint8_t *dst,
                    size_t dst_stride, int n_planes, size_t src_ps, size_t dst_ps, int where) try {
    if (!c || !src || !dst || cw < 1 || ch < 1 || n_planes < 0) return HVC_E_INVALID_ARG;
    if (where != HVC_MEM_HOST && where != HVC_MEM_DEVICE) return HVC_E_INVALID_ARG;
    if (src_stride < (size_t)cw || dst_stride < (size_t)cw * 2) return HVC_E_INVALID_ARG;
    if (n_planes == 0) return HVC_OK;
    if (n_planes > 65535) return HVC_E_TOO_LARGE;
    if (!src_ps) src_ps = src_stride * (size_t)ch;
    if (!dst_ps) dst_ps = dst_stride * (size_t)ch * 2;
    DeviceGuard g(c->device);
    if (!g.ok) return fail_hip(c, hipErrorInvalidDevice);
    hvc::UpsampleParams P;
    std::memset(&P, 0, sizeof P);
    P.cw = cw;
    P.ch = ch;
    P.n_planes = n_planes;
    P.src_stride = src_stride;
    P.dst_stride = dst_stride;
    P.src_ps = src_ps;
    P.dst_ps = dst_ps;
    if (where == HVC_MEM_DEVICE) {
        P.src = src;
        P.dst = dst;
        HIPCHK(c, hvc::launch_upsample420(P, c->stream));
        return HVC_OK;
    }
    size_t sbytes = (size_t)(n_planes - 1) * src_ps + (size_t)(ch - 1) * src_stride + (size_t)cw;
    size_t dbytes = (size_t)(n_planes - 1) * dst_ps + (size_t)(2 * ch - 1) * dst_stride + (size_t)cw * 2;
    int r = grow(c, &c->d_in, &c->in_cap, sbytes);
    if (r) return r;
    r = grow(c, &c->d_out, &c->out_cap, dbytes);
    if (r) return r;
    HIPCHK(c, hipMemcpyAsync(c->d_in, src, sbytes, hipMemcpyHostToDevice, c->stream));
    P.src = (const uint8_t *)c->d_in;
    P.dst = (uint8_t *)c->d_out;
    HIPCHK(c, hvc::launch_upsample420(P, c->stream));
    for (int p = 0; p < n_planes; p++)
        HIPCHK(c, hipMemcpy2DAsync(dst + (size_t)p * dst_ps, dst_stride, (uint8_t *)c->d_out + (size_t)p * dst_ps,
                                   dst_stride, (size_t)cw * 2, (size_t)ch * 2, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return HVC_OK;
} HVC_ABI_CATCH

// ---------------------------------------------------------------------------
// single-frame conveniences (host memory)

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
static int gpu_entropy_decode(hvc_ctx *c, const uint8_t *const *jpegs, const size_t *sizes, int n_frames,
                              const hvc_jpeg_info &info0, int16_t *d_coefs, size_t coef_fs, int *used_gpu,
                              AfterReader *after = nullptr);

// One file: Huffman reader on the GPU (hvc_hdec.hip) into device scratch; *used = 0 when the stream needs the
// host decoder (nothing usable on the device then).
static int single_frame_coefs_on_device(hvc_ctx *c, const uint8_t *jpeg, size_t n, const hvc_jpeg_info *info, int *used,
                                        AfterReader *after = nullptr) {
    *used = 0;
    // Below ~128 kB the host reader is done before the GPU decoder's launches and synchronisations are
    // (tools/bench_single.py on 1080p: 64 kB file 0.39 ms on the host vs 0.8 ms; 228 kB 1.7 vs 0.8 ms; 967 kB 3.9 vs 1.6 ms).
    if (n < 128u * 1024u) return HVC_OK;
    DeviceGuard g(c->device);
    if (!g.ok) return fail_hip(c, hipErrorInvalidDevice);
    int r = grow(c, &c->gd_coefs, &c->gd_coefs_cap, info->coef_count * sizeof(int16_t));
    if (r) return r;
    if (after && c->decode_kernel != 1 && c->decode_kernel != 3) { // (the A/B alternates read the DC from the record)
        const size_t blocks = info->coef_count / 64;
        if ((r = grow(c, &c->gd_dcv, &c->gd_dcv_cap, ((blocks + 127) & ~(size_t)127) * sizeof(int16_t)))) return r;
        after->dc_plane = (int16_t *)c->gd_dcv;
        after->dc_fs = blocks;
    }
    return gpu_entropy_decode(c, &jpeg, &n, 1, *info, (int16_t *)c->gd_coefs, info->coef_count, used, after);
}

// One frame whose record came from the host reader with blocks on the wide-DC list: upload, block stage, the int64
// fix-up with the true DCs, download -- the model's output for a stream whose DC leaves int16 (decoder.ml:143).
static int decode_one_with_wide_dc(hvc_ctx *c, const hvc_jpeg_info *info, const int16_t *coefs, const std::vector<hvc::WideDc> &wide,
                                   bool yuv444, uint8_t *out) {
    DeviceGuard g(c->device);
    if (!g.ok) return fail_hip(c, hipErrorInvalidDevice);
    const size_t cb = info->coef_count * sizeof(int16_t);
    const size_t ob = yuv444 ? (size_t)3 * info->width * info->height : info->pixel_bytes;
    int r;
    if ((r = grow(c, &c->d_in, &c->in_cap, cb))) return r;
    if ((r = grow(c, &c->d_out, &c->out_cap, ob))) return r;
    std::vector<WideFix> fix;
    try {
        for (const hvc::WideDc &w : wide) fix.push_back(WideFix{0, w.block, w.dc});
    } catch (const std::bad_alloc &) {
        return HVC_E_OUT_OF_MEMORY;
    }
    HIPCHK(c, hipMemcpyAsync(c->d_in, coefs, cb, hipMemcpyHostToDevice, c->stream));
    const bool prof_saved = c->profiling;
    c->profiling = false;
    r = yuv444 ? decode_frames_yuv444_impl(c, (const int16_t *)c->d_in, info->coef_count, &info->qtabs[0][0], info->n_qtabs,
                                           info->layout, info->n_comp, 1, info->width, info->height, (uint8_t *)c->d_out, ob,
                                           HVC_MEM_DEVICE, nullptr, 0, &fix)
               : decode_frames_impl(c, (const int16_t *)c->d_in, info->coef_count, &info->qtabs[0][0], info->n_qtabs, info->layout,
                                    info->n_comp, 1, (uint8_t *)c->d_out, info->pixel_bytes, HVC_MEM_DEVICE, nullptr, 0, &fix);
    c->profiling = prof_saved;
    if (r) return r;
    HIPCHK(c, hipMemcpyAsync(out, c->d_out, ob, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return HVC_OK;
}

// Decoder.decode_a_frame minus the crop (decoder.ml:422-427)
int hvc_jpeg_decode_yuv444(hvc_ctx *c, const uint8_t *jpeg, size_t n, hvc_jpeg_info *info, uint8_t *frame,
                           size_t frame_cap) try {
    if (!c || !jpeg || !info || !frame) return HVC_E_INVALID_ARG;
    int r = hvc_jpeg_read_header(jpeg, n, info);
    if (r) return r;
    // a 4:2:0 scan: Y 2x2, Cb / Cr 1x1 (Frame.infer_chroma_subsampling, common/src/frame.ml:42-61)
    if (info->n_comp != 3 || info->comp[0].hscale != 2 || info->comp[0].vscale != 2 || info->comp[1].hscale != 1 ||
        info->comp[1].vscale != 1 || info->comp[2].hscale != 1 || info->comp[2].vscale != 1)
        return HVC_E_INVALID_ARG;
    if (frame_cap < (size_t)3 * info->width * info->height) return HVC_E_INVALID_ARG;
    int on_gpu = 0;
    const size_t fb = (size_t)3 * info->width * info->height;
    AfterReader after;
    // coefficient record on the device: fused block stage there, one download -- enqueued behind the reader at once
    auto block_stage = [&]() -> int {
        DeviceGuard g(c->device);
        int e = grow(c, &c->d_out, &c->out_cap, fb);
        if (e) return e;
        e = decode_frames_yuv444_impl(c, (const int16_t *)c->gd_coefs, info->coef_count, &info->qtabs[0][0], info->n_qtabs,
                                      info->layout, info->n_comp, 1, info->width, info->height, (uint8_t *)c->d_out, fb,
                                      HVC_MEM_DEVICE, after.dc_plane, after.dc_fs);
        if (e) return e;
        HIPCHK(c, hipMemcpyAsync(frame, c->d_out, fb, hipMemcpyDeviceToHost, c->stream));
        return HVC_OK;
    };
    after.enqueue = block_stage;
    if ((r = single_frame_coefs_on_device(c, jpeg, n, info, &on_gpu, &after))) return r;
    if (on_gpu) {
        if (after.speculated) return HVC_OK; // (the reader's one synchronisation covered the download)
        DeviceGuard g(c->device);
        if ((r = block_stage())) return r;
        HIPCHK(c, hipStreamSynchronize(c->stream));
        return HVC_OK;
    }
    std::vector<int16_t> coefs;
    try {
        coefs.resize(info->coef_count);
    } catch (const std::bad_alloc &) {
        return HVC_E_OUT_OF_MEMORY;
    }
    std::vector<hvc::WideDc> wide;
    r = hvc::entropy_decode_wide(jpeg, n, info, coefs.data(), wide);
    if (r) return r;
    if (!wide.empty()) return decode_one_with_wide_dc(c, info, coefs.data(), wide, true, frame);
    return hvc_decode_frames_yuv444(c, coefs.data(), info->coef_count, &info->qtabs[0][0], info->n_qtabs, info->layout,
                                    info->n_comp, 1, info->width, info->height, frame,
                                    (size_t)3 * info->width * info->height, HVC_MEM_HOST);
} HVC_ABI_CATCH

int hvc_jpeg_decode(hvc_ctx *c, const uint8_t *jpeg, size_t n, hvc_jpeg_info *info, uint8_t *pixels, size_t pixel_cap) try {
    if (!c || !jpeg || !info || !pixels) return HVC_E_INVALID_ARG;
    int r = hvc_jpeg_read_header(jpeg, n, info);
    if (r) return r;
    if (pixel_cap < info->pixel_bytes) return HVC_E_INVALID_ARG;
    int on_gpu = 0;
    AfterReader after;
    // coefficient record on the device: ALL components' block stage there in one launch, one download -- enqueued
    // behind the reader at once
    auto block_stage = [&]() -> int {
        DeviceGuard g(c->device);
        int e = grow(c, &c->d_out, &c->out_cap, info->pixel_bytes);
        if (e) return e;
        e = decode_frames_impl(c, (const int16_t *)c->gd_coefs, info->coef_count, &info->qtabs[0][0], info->n_qtabs, info->layout,
                               info->n_comp, 1, (uint8_t *)c->d_out, info->pixel_bytes, HVC_MEM_DEVICE, after.dc_plane, after.dc_fs);
        if (e) return e;
        HIPCHK(c, hipMemcpyAsync(pixels, c->d_out, info->pixel_bytes, hipMemcpyDeviceToHost, c->stream));
        return HVC_OK;
    };
    after.enqueue = block_stage;
    if ((r = single_frame_coefs_on_device(c, jpeg, n, info, &on_gpu, &after))) return r;
    if (on_gpu) {
        if (after.speculated) return HVC_OK; // (the reader's one synchronisation covered the download)
        DeviceGuard g(c->device);
        if ((r = block_stage())) return r;
        HIPCHK(c, hipStreamSynchronize(c->stream));
        return HVC_OK;
    }
    std::vector<int16_t> coefs;
    try {
        coefs.resize(info->coef_count);
    } catch (const std::bad_alloc &) {
        return HVC_E_OUT_OF_MEMORY;
    }
    std::vector<hvc::WideDc> wide;
    r = hvc::entropy_decode_wide(jpeg, n, info, coefs.data(), wide);
    if (r) return r;
    if (!wide.empty()) return decode_one_with_wide_dc(c, info, coefs.data(), wide, false, pixels);
    return hvc_decode_frames(c, coefs.data(), info->coef_count, &info->qtabs[0][0], info->n_qtabs, info->layout,
                             info->n_comp, 1, pixels, info->pixel_bytes, HVC_MEM_HOST);
} HVC_ABI_CATCH

// Encoder.encode_420/422/444 (encoder.ml:512-541)
static int huffman_prepare(hvc_ctx *c, const hvc_jpeg_info *info, const int16_t *d_coefs, size_t coef_fs, int n_frames,
                           uint8_t *d_out, size_t out_cap, unsigned long long *d_offsets, hvc::HuffParams &P);

int hvc_jpeg_encode(hvc_ctx *c, const uint8_t *y, const uint8_t *u, const uint8_t *v, int width, int height, int chroma,
                    int quality, uint8_t *out, size_t cap, size_t *out_len) try {
    if (!c || !y || !u || !v || !out_len) return HVC_E_INVALID_ARG;
    hvc_jpeg_info info;
    int r = hvc_jpeg_encoder_layout(width, height, chroma, quality, &info);
    if (r) return r;
    if ((r = hvc_jpeg_encoder_check(&info))) return r; // the model raises for this geometry
    std::vector<uint8_t> planes, header;
    try {
        planes.assign(info.pixel_bytes, 0); // Plane.create is zero-filled (plane.ml:11-17)
    } catch (const std::bad_alloc &) {
        return HVC_E_OUT_OF_MEMORY;
    }
    // Plane.blit_available of the frame's planes into the padded ones (encoder.ml:514-516; frame.ml:10-41)
    const uint8_t *src[3] = {y, u, v};
    const int cw = chroma == 444 ? width : width / 2, ch = chroma == 420 ? height / 2 : height;
    const int sw[3] = {width, cw, cw}, sh[3] = {height, ch, ch};
    for (int i = 0; i < 3; i++) {
        const int bw = sw[i] < info.comp[i].decoded_width ? sw[i] : info.comp[i].decoded_width;
        const int bh = sh[i] < info.comp[i].decoded_height ? sh[i] : info.comp[i].decoded_height;
        for (int row = 0; row < bh; row++)
            std::memcpy(planes.data() + info.layout[i].plane_offset + (size_t)row * info.layout[i].stride,
                        src[i] + (size_t)row * sw[i], (size_t)bw);
    }
    // forward block stage and Huffman coder both on the device; only the entropy-coded segment comes back
    DeviceGuard g(c->device);
    if (!g.ok) return fail_hip(c, hipErrorInvalidDevice);
    const size_t coef_bytes = info.coef_count * sizeof(int16_t);
    const size_t seg_cap = (info.coef_count / 64) * 243 + 64; // worst case incl. stuffing
    if ((r = grow(c, &c->d_in, &c->in_cap, info.pixel_bytes))) return r;
    if ((r = grow(c, &c->d_out, &c->out_cap, coef_bytes))) return r;
    if ((r = grow(c, &c->hd_out, &c->hd_out_cap, seg_cap))) return r;
    HIPCHK(c, hipMemcpyAsync(c->d_in, planes.data(), info.pixel_bytes, hipMemcpyHostToDevice, c->stream));
    const bool prof_saved = c->profiling;
    c->profiling = false;
    r = hvc_encode_frames(c, (const uint8_t *)c->d_in, info.pixel_bytes, &info.qtabs[0][0], info.n_qtabs, info.layout, 3, 1,
                          (int16_t *)c->d_out, info.coef_count, HVC_MEM_DEVICE);
    c->profiling = prof_saved;
    if (r) return r;
    hvc::HuffParams P;
    if ((r = huffman_prepare(c, &info, (const int16_t *)c->d_out, info.coef_count, 1, (uint8_t *)c->hd_out, seg_cap, nullptr, P)))
        return r;
    HIPCHK(c, hvc::launch_huffman_encode(P, c->stream));
    unsigned status = 0;
    unsigned long long off[2] = {0, 0};
    HIPCHK(c, hipMemcpyAsync(&status, P.status, sizeof status, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(off, P.out_offsets, sizeof off, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (status & 1u) return HVC_E_RANGE;
    if ((status & 6u) || off[1] > seg_cap) return HVC_E_TOO_LARGE;
    hvc::jpeg_header_bytes(&info, header);
    *out_len = header.size() + (size_t)off[1] + 2;
    if (!out || *out_len > cap) return HVC_E_INVALID_ARG;
    std::memcpy(out, header.data(), header.size());
    HIPCHK(c, hipMemcpy(out + header.size(), c->hd_out, (size_t)off[1], hipMemcpyDeviceToHost));
    out[header.size() + off[1]] = 0xff; // complete_and_write_eoi (encoder.ml:507-510)
    out[header.size() + off[1] + 1] = 0xd9;
    return HVC_OK;
} HVC_ABI_CATCH

// ---------------------------------------------------------------------------
// BASELINE config 3: host Huffman || hipMemcpyAsync (copy stream) || block-stage kernel (compute stream)
static bool is_420_scan(const hvc_jpeg_info &info) { // Y 2x2, Cb / Cr 1x1 (frame.ml:42-61)
    return info.n_comp == 3 && info.comp[0].hscale == 2 && info.comp[0].vscale == 2 && info.comp[1].hscale == 1 &&
           info.comp[1].vscale == 1 && info.comp[2].hscale == 1 && info.comp[2].vscale == 1;
}

// yuv444 = false: padded component planes per frame (hvc_jpeg_decode_batch);
// yuv444 = true: tight 4:4:4 frames through the fused kernel (hvc_jpeg_decode_batch_yuv444)
static int decode_batch_impl(hvc_ctx *c, const uint8_t *const *jpegs, const size_t *sizes, int n_frames, int threads,
                             int frames_per_chunk, uint8_t *pixels, size_t pixel_fs, int where, hvc_batch_stats *stats,
                             bool yuv444) {
    if (!c || !jpegs || !sizes || !pixels || n_frames < 0) return HVC_E_INVALID_ARG;
    if (where != HVC_MEM_HOST && where != HVC_MEM_DEVICE) return HVC_E_INVALID_ARG;
    if (stats) std::memset(stats, 0, sizeof *stats);
    if (n_frames == 0) return HVC_OK;
    hvc_jpeg_info info0;
    int r = hvc_jpeg_read_header(jpegs[0], sizes[0], &info0);
    if (r) return r;
    if (yuv444 && (!is_420_scan(info0) || (info0.width & 1) || (info0.height & 1))) return HVC_E_INVALID_ARG;
    const size_t out_bytes = yuv444 ? (size_t)3 * info0.width * info0.height : info0.pixel_bytes; // per frame
    if (pixel_fs < out_bytes || (!yuv444 && (pixel_fs & 7))) return HVC_E_INVALID_ARG;
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    if (frames_per_chunk < 1) frames_per_chunk = 32;
    if (frames_per_chunk > n_frames) frames_per_chunk = n_frames;
    const int C = frames_per_chunk, NB = hvc_ctx::RING;
    const int n_chunks = (n_frames + C - 1) / C;
    const size_t frame_coef_bytes = info0.coef_count * sizeof(int16_t);
    const size_t ring_bytes = frame_coef_bytes * (size_t)C;
    const size_t oring_bytes = where == HVC_MEM_HOST ? out_bytes * (size_t)C : 0;

    DeviceGuard g(c->device);
    if (!g.ok) return fail_hip(c, hipErrorInvalidDevice);
    if (!c->copy_stream) HIPCHK(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    for (int i = 0; i < NB; i++) {
        if (!c->ev_h2d[i]) HIPCHK(c, hipEventCreate(&c->ev_h2d[i]));
        if (!c->ev_kern[i]) HIPCHK(c, hipEventCreate(&c->ev_kern[i]));
    }
    for (int i = 0; i < 4; i++)
        if (!c->ev_t[i]) HIPCHK(c, hipEventCreate(&c->ev_t[i]));
    if (ring_bytes > c->ring_bytes) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, hipStreamSynchronize(c->copy_stream));
        for (int i = 0; i < NB; i++) {
            if (c->h_ring[i]) (void)hipHostFree(c->h_ring[i]);
            if (c->d_ring[i]) (void)hipFree(c->d_ring[i]);
            c->h_ring[i] = c->d_ring[i] = nullptr;
        }
        c->ring_bytes = 0;
        for (int i = 0; i < NB; i++) {
            if (hipHostMalloc(&c->h_ring[i], ring_bytes, hipHostMallocDefault) != hipSuccess ||
                hipMalloc(&c->d_ring[i], ring_bytes) != hipSuccess)
                return HVC_E_OUT_OF_MEMORY;
        }
        c->ring_bytes = ring_bytes;
    }
    if (oring_bytes > c->oring_bytes) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        for (int i = 0; i < NB; i++) {
            if (c->d_oring[i]) (void)hipFree(c->d_oring[i]);
            c->d_oring[i] = nullptr;
        }
        c->oring_bytes = 0;
        for (int i = 0; i < NB; i++)
            if (hipMalloc(&c->d_oring[i], oring_bytes) != hipSuccess) return HVC_E_OUT_OF_MEMORY;
        c->oring_bytes = oring_bytes;
    }

    // worker threads pull frames in order; a frame's chunk slot must have been released (its previous
    // occupant uploaded) before they write into it
    std::mutex mu;
    std::condition_variable cv;
    std::atomic<int> next_frame{0};
    std::atomic<int> error{0};
    std::vector<int> done_in_chunk((size_t)n_chunks, 0);
    std::vector<std::vector<WideFix>> chunk_wide((size_t)n_chunks); // blocks whose DC left int16 (frame = index in the chunk)
    int released_upto = NB - 1; // chunks 0..NB-1 may be written at once
    std::atomic<long long> entropy_ns{0};
    auto worker_body = [&]() {
        if (!pin_to_ctx_cpus(c)) error.store(HVC_E_INVALID_ARG); // hvc_set_host_cpus
        // Frames are taken TWO at a time and decoded symbol by symbol in turn (hvc::entropy_decode_wide2): one stream is
        // one dependency chain, two streams are two chains the core overlaps -- 1.4x the frames per second per thread.
        std::vector<hvc::WideDc> wide2[2];
        static const int take = [] { const char *v = std::getenv("HVC_HOST_PAIRS"); return v && v[0] == '0' ? 1 : 2; }(); // (A/B: 0 = one file at a time)
        for (;;) {
            const int f0 = next_frame.fetch_add(take);
            if (f0 >= n_frames || error.load()) return;
            const int cnt = (take == 2 && f0 + 1 < n_frames) ? 2 : 1;
            const int k_last = (f0 + cnt - 1) / C;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return k_last <= released_upto || error.load(); });
            }
            if (error.load()) return;
            const auto t0 = std::chrono::steady_clock::now();
            hvc_jpeg_info fi[2];
            int e[2] = {HVC_OK, HVC_OK};
            int16_t *dst[2] = {nullptr, nullptr};
            for (int q = 0; q < cnt; q++) {
                const int f = f0 + q, k = f / C;
                e[q] = hvc_jpeg_read_header(jpegs[f], sizes[f], &fi[q]);
                if (!e[q] && (fi[q].n_comp != info0.n_comp || fi[q].n_qtabs != info0.n_qtabs || fi[q].coef_count != info0.coef_count ||
                              std::memcmp(fi[q].layout, info0.layout, sizeof fi[q].layout) ||
                              std::memcmp(fi[q].qtabs, info0.qtabs, sizeof fi[q].qtabs)))
                    e[q] = HVC_E_INVALID_ARG; // a batch shares one geometry and one set of tables
                dst[q] = (int16_t *)c->h_ring[k % NB] + (size_t)(f - k * C) * info0.coef_count;
            }
            if (cnt == 2 && !e[0] && !e[1]) {
                const uint8_t *const data[2] = {jpegs[f0], jpegs[f0 + 1]};
                const size_t len[2] = {sizes[f0], sizes[f0 + 1]};
                const hvc_jpeg_info *const inf[2] = {&fi[0], &fi[1]};
                std::vector<hvc::WideDc> *const wd[2] = {&wide2[0], &wide2[1]};
                hvc::entropy_decode_wide2(data, len, inf, dst, wd, e);
            } else {
                for (int q = 0; q < cnt; q++)
                    if (!e[q]) e[q] = hvc::entropy_decode_wide(jpegs[f0 + q], sizes[f0 + q], &fi[q], dst[q], wide2[q]);
            }
            entropy_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
            std::lock_guard<std::mutex> lk(mu);
            for (int q = 0; q < cnt; q++) {
                const int f = f0 + q, k = f / C;
                if (!e[q] && !wide2[q].empty()) {
                    try {
                        for (const hvc::WideDc &w : wide2[q]) chunk_wide[(size_t)k].push_back(WideFix{f - k * C, w.block, w.dc});
                    } catch (const std::bad_alloc &) {
                        e[q] = HVC_E_OUT_OF_MEMORY;
                    }
                }
                if (e[q] && !error.load()) error.store(e[q]); // (the pair's first error: frame order)
                done_in_chunk[(size_t)k]++;
            }
            cv.notify_all();
        }
    };
    auto worker = [&]() { // (a pool thread: nothing may leave it but through the error flag the orchestrator watches)
        try {
            worker_body();
        } catch (...) {
            const int e = hvc::exception_code();
            std::lock_guard<std::mutex> lk(mu);
            error.store(e);
            cv.notify_all();
        }
    };
    const auto wall0 = std::chrono::steady_clock::now();
    if ((r = pool_ready(c, threads))) return r;
    bool completed = false; // (the workers have run out of frames by themselves)
    hvc::PoolScope scope(c->pool, [&] {
        std::lock_guard<std::mutex> lk(mu);
        if (!completed && !error.load()) error.store(HVC_E_INTERNAL);
        cv.notify_all();
    });
    if ((r = c->pool.submit(worker, threads))) {
        std::lock_guard<std::mutex> lk(mu);
        error.store(r);
        return r; // (the scope waits for the copies that were queued)
    }

    int rc = HVC_OK;
    double h2d_ms = 0, k_ms = 0, d2h_ms = 0;
    hipStream_t compute = c->stream;
    try {
    for (int k = 0; k < n_chunks && rc == HVC_OK; k++) {
        const int slot = k % NB, first = k * C, cnt = (first + C <= n_frames) ? C : n_frames - first;
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return done_in_chunk[(size_t)k] == cnt || error.load(); });
        }
        if (error.load()) { rc = error.load(); break; }
        hipError_t he = hipSuccess;
        uint8_t *dst = where == HVC_MEM_DEVICE ? pixels + (size_t)first * pixel_fs : (uint8_t *)c->d_oring[slot];
        const size_t dst_fs = where == HVC_MEM_DEVICE ? pixel_fs : out_bytes;
        // the device chunk (and output ring slot) is reused every NB chunks: its previous kernel must be done
        if (k >= NB) he = hipStreamWaitEvent(c->copy_stream, c->ev_kern[slot], 0);
        if (he == hipSuccess) he = hipEventRecord(c->ev_t[0], c->copy_stream);
        if (he == hipSuccess)
            he = hipMemcpyAsync(c->d_ring[slot], c->h_ring[slot], frame_coef_bytes * (size_t)cnt, hipMemcpyHostToDevice,
                                c->copy_stream);
        if (he == hipSuccess) he = hipEventRecord(c->ev_h2d[slot], c->copy_stream);
        if (he == hipSuccess) he = hipStreamWaitEvent(compute, c->ev_h2d[slot], 0);
        if (he != hipSuccess) { rc = fail_hip(c, he); break; }
        const bool prof_saved = c->profiling;
        c->profiling = false;
        he = hipEventRecord(c->ev_t[1], compute);
        const std::vector<WideFix> *wf = &chunk_wide[(size_t)k]; // (complete: the chunk's workers are done)
        rc = yuv444 ? decode_frames_yuv444_impl(c, (const int16_t *)c->d_ring[slot], info0.coef_count, &info0.qtabs[0][0],
                                                info0.n_qtabs, info0.layout, info0.n_comp, cnt, info0.width, info0.height,
                                                dst, dst_fs, HVC_MEM_DEVICE, nullptr, 0, wf)
                    : decode_frames_impl(c, (const int16_t *)c->d_ring[slot], info0.coef_count, &info0.qtabs[0][0],
                                         info0.n_qtabs, info0.layout, info0.n_comp, cnt, dst, dst_fs, HVC_MEM_DEVICE, nullptr, 0, wf);
        c->profiling = prof_saved;
        if (rc) break;
        if (he == hipSuccess) he = hipEventRecord(c->ev_t[2], compute);
        if (he == hipSuccess && where == HVC_MEM_HOST) {
            for (int f = 0; f < cnt && he == hipSuccess; f++)
                he = hipMemcpyAsync(pixels + (size_t)(first + f) * pixel_fs, dst + (size_t)f * dst_fs, out_bytes,
                                    hipMemcpyDeviceToHost, compute);
        }
        if (he == hipSuccess) he = hipEventRecord(c->ev_kern[slot], compute);
        if (he == hipSuccess) he = hipEventRecord(c->ev_t[3], compute);
        // wait for this chunk's upload, then hand the pinned slot to chunk k + NB
        if (he == hipSuccess) he = wait_event(c->ev_h2d[slot]);
        if (he != hipSuccess) { rc = fail_hip(c, he); break; }
        {
            std::lock_guard<std::mutex> lk(mu);
            released_upto = k + NB;
            cv.notify_all();
        }
        float ms = 0;
        if (hipEventElapsedTime(&ms, c->ev_t[0], c->ev_h2d[slot]) == hipSuccess) h2d_ms += ms;
        // kernel / d2h times of this chunk: the events are shared by all chunks, so they are read (and the
        // chunk waited for) before the next one records them; the worker threads -- the bound of this
        // pipeline -- keep decoding into the other ring slots meanwhile
        if (wait_event(c->ev_t[3]) == hipSuccess) {
            if (hipEventElapsedTime(&ms, c->ev_t[1], c->ev_t[2]) == hipSuccess) k_ms += ms;
            if (hipEventElapsedTime(&ms, c->ev_t[2], c->ev_t[3]) == hipSuccess) d2h_ms += ms;
        }
    }
    } catch (...) {
        rc = hvc::exception_code();
    }
    {
        std::lock_guard<std::mutex> lk(mu);
        if (rc != HVC_OK) error.store(rc);
        else completed = true;
        cv.notify_all();
    }
    {
        const int te = scope.finish();
        if (rc == HVC_OK && te) rc = te;
    }
    if (rc == HVC_OK && error.load()) rc = error.load();
    if (rc == HVC_OK) {
        hipError_t he = hipStreamSynchronize(compute);
        if (he == hipSuccess) he = hipStreamSynchronize(c->copy_stream);
        if (he != hipSuccess) rc = fail_hip(c, he);
    } else {
        (void)hipStreamSynchronize(compute);
        (void)hipStreamSynchronize(c->copy_stream);
    }
    if (stats) {
        stats->wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
        stats->entropy_ms_sum = (double)entropy_ns.load() * 1e-6;
        stats->h2d_ms_sum = h2d_ms;
        stats->kernel_ms_sum = k_ms;
        stats->d2h_ms_sum = d2h_ms;
        stats->chunks = n_chunks;
        stats->threads = threads;
        stats->frames_per_chunk = C;
        stats->coef_bytes = (uint64_t)frame_coef_bytes * (uint64_t)n_frames;
    }
    return rc;
}

int hvc_jpeg_decode_batch(hvc_ctx *c, const uint8_t *const *jpegs, const size_t *sizes, int n_frames, int threads,
                          int frames_per_chunk, uint8_t *pixels, size_t pixel_fs, int where, hvc_batch_stats *stats) try {
    return decode_batch_impl(c, jpegs, sizes, n_frames, threads, frames_per_chunk, pixels, pixel_fs, where, stats, false);
} HVC_ABI_CATCH

int hvc_jpeg_decode_batch_yuv444(hvc_ctx *c, const uint8_t *const *jpegs, const size_t *sizes, int n_frames,
                                 int threads, int frames_per_chunk, uint8_t *frames, size_t frame_stride, int where,
                                 hvc_batch_stats *stats) try {
    return decode_batch_impl(c, jpegs, sizes, n_frames, threads, frames_per_chunk, frames, frame_stride, where, stats,
                             true);
} HVC_ABI_CATCH

// Two files of one batch: the same frame geometry (sizes, sampling, planes)?  The Huffman table selectors of the scan
// may differ -- the GPU reader takes every file's tables from the file itself.
static bool same_geometry(const hvc_jpeg_info &a, const hvc_jpeg_info &b) {
    if (a.n_comp != b.n_comp || a.coef_count != b.coef_count || a.width != b.width || a.height != b.height ||
        std::memcmp(a.layout, b.layout, sizeof a.layout))
        return false;
    for (int i = 0; i < a.n_comp; i++) {
        const hvc_jpeg_component &x = a.comp[i], &y = b.comp[i];
        if (x.identifier != y.identifier || x.hscale != y.hscale || x.vscale != y.vscale || x.decoded_width != y.decoded_width ||
            x.decoded_height != y.decoded_height || x.actual_width != y.actual_width || x.actual_height != y.actual_height)
            return false;
    }
    return true;
}

// Geometry part of the GPU Huffman decoder's parameter block; false = this frame layout needs the host decoder.
static bool gd_geometry(const hvc_jpeg_info &info0, hvc::HdParams &P) {
    std::memset(&P, 0, sizeof P);
    if (info0.n_comp < 1 || info0.n_comp > 3) return false;
    P.n_comp = info0.n_comp;
    const hvc_jpeg_component &c0 = info0.comp[0];
    if (c0.hscale < 1 || c0.vscale < 1) return false;
    P.mbs_wide = c0.decoded_width / (8 * c0.hscale);
    P.mbs_high = c0.decoded_height / (8 * c0.vscale);
    int base = 0;
    for (int i = 0; i < info0.n_comp; i++) {
        P.comp[i].h = info0.comp[i].hscale;
        P.comp[i].v = info0.comp[i].vscale;
        P.comp[i].bw = info0.layout[i].blocks_w;
        P.comp[i].mcu_base = base;
        P.comp[i].coef_off = info0.layout[i].coef_offset;
        if (P.comp[i].h < 1 || P.comp[i].v < 1) return false;
        // the decoder raises when the MCU grid leaves a plane ("Plane.set out of bounds"): host path decides
        if (P.mbs_wide * P.comp[i].h > info0.layout[i].blocks_w || P.mbs_high * P.comp[i].v > info0.layout[i].blocks_h)
            return false;
        if (base + P.comp[i].h * P.comp[i].v > HVC_HD_MAX_MCU_BLOCKS) return false;
        for (int k = 0; k < P.comp[i].h * P.comp[i].v; k++) P.b2comp[base + k] = (unsigned char)i;
        base += P.comp[i].h * P.comp[i].v;
    }
    P.blocks_per_mcu = base;
    const unsigned long long bpf = (unsigned long long)P.mbs_wide * P.mbs_high * base;
    if (bpf == 0 || bpf >= (1ull << 31) || info0.coef_count >= (1ull << 32)) return false;
    P.blocks_per_frame = (unsigned)bpf;
    return true;
}

// The per-subsequence arrays of the GPU Huffman decoder inside one allocation of HVC_HD_STATE_BYTES(n).
static void gd_carve_state(hvc::HdParams &P, void *mem, size_t n) {
    unsigned long long *sp = (unsigned long long *)mem;
    P.start_used = sp;
    P.exit_a = sp + n;
    P.exit_b = sp + 2 * n;
    P.exit_c = sp + 3 * n;
    unsigned *up = (unsigned *)(sp + 4 * n);
    P.nblk = up;
    P.list0 = up + n;
    P.list1 = up + 2 * n;
    P.list_n = up + 3 * n;
}

// Huffman tables of a batch -> device: the value tables and, when the components use at most two table
// sets, the synchronisation tables (HdSpec) behind them.  Fills P.tables / P.spec / P.slotmask.
static int gd_upload_tables(hvc_ctx *c, const hvc::HdTables &t, hvc::HdParams &P, hipStream_t st) {
    int r;
    if ((r = grow(c, &c->gd_tables, &c->gd_tables_cap, sizeof(hvc::HdTables) + sizeof(hvc::HdSpec) + sizeof(hvc::HdSpecOvf)))) return r;
    hvc::HdSpec spec;
    hvc::HdSpecOvf spec_ovf; // (the overflow records of tables with more than HVC_HD_SUBTABLES long prefixes: hvc_hdec.h)
    unsigned char slot[4];
    static const bool classic = std::getenv("HVC_HD_CLASSIC") != nullptr; // tests: force k_hd_round / k_hd_write
    const bool have_spec = hvc::make_spec(t, P.n_comp, spec, slot, P.slot_rep, &spec_ovf) && !classic;
    // file after file with the same tables (the usual case: an encoder's fixed set) finds them on the device already
    if (!c->gd_tables_host) c->gd_tables_host = new (std::nothrow) hvc::HdTables;
    if (!c->gd_tables_host) return HVC_E_OUT_OF_MEMORY;
    if (!(c->gd_tables_valid && c->gd_tables_ncomp == P.n_comp && !std::memcmp(c->gd_tables_host, &t, sizeof t))) {
        c->gd_tables_valid = false;
        HIPCHK(c, hipMemcpyAsync(c->gd_tables, &t, sizeof t, hipMemcpyHostToDevice, st));
        if (have_spec) {
            HIPCHK(c, hipMemcpyAsync((char *)c->gd_tables + sizeof t, &spec, sizeof spec, hipMemcpyHostToDevice, st));
            HIPCHK(c, hipMemcpyAsync((char *)c->gd_tables + sizeof t + sizeof spec, &spec_ovf, sizeof spec_ovf, hipMemcpyHostToDevice, st));
        }
        HIPCHK(c, hipStreamSynchronize(st)); // the sources live on a stack frame
        std::memcpy(c->gd_tables_host, &t, sizeof t);
        c->gd_tables_ncomp = P.n_comp;
        c->gd_tables_valid = true;
    }
    P.tables = (const hvc::HdTables *)c->gd_tables;
    P.spec = have_spec ? (const hvc::HdSpec *)((char *)c->gd_tables + sizeof t) : nullptr;
    P.spec_ovf = have_spec ? (const hvc::HdSpecOvf *)((char *)c->gd_tables + sizeof t + sizeof(hvc::HdSpec)) : nullptr;
    P.ftabs = nullptr;
    P.tabset_of = nullptr;
    P.slotmask = P.selmask = 0;
    for (int b = 0; b < P.blocks_per_mcu; b++) {
        P.slotmask |= (unsigned)slot[P.b2comp[b]] << b;
        P.selmask |= (unsigned)slot[P.b2comp[b]] << (2 * b);
    }
    return HVC_OK;
}

// PF mode: one work list per frame (hvc::HdParams::list_fn) pays where a frame fills workgroups of 512 subsequences by
// itself -- 1080p files have 7 000 -- and the frames fit the launch grid's second dimension; batches of small files
// keep the batch-wide lists, which pack the subsequences of many frames into one workgroup.
static bool gd_lists_per_frame(size_t total_sub, int n_frames) {
    return n_frames >= 1 && n_frames <= 65535 && total_sub / (size_t)n_frames >= 1024;
}

// PF mode (per-frame Huffman tables, hvc_hdec.h): which tables block b of an MCU reads = its component
static unsigned gd_component_selmask(const hvc::HdParams &P) {
    unsigned m = 0;
    for (int b = 0; b < P.blocks_per_mcu; b++) m |= (unsigned)P.b2comp[b] << (2 * b);
    return m;
}

// Enqueue the whole decode on `st`: the clearing launch (frame_of, flags, list lengths), `rounds` synchronisation
// launches, the finish passes -- kernels only, no memset node in between.  Afterwards *P.changed holds the number of the
// last launch that still changed something (gd_unsettled), *P.status the error bits.
static hipError_t gd_enqueue(const hvc::HdParams &P, int rounds, hipStream_t st) {
    hipError_t e = hvc::launch_hd_frame_of(P, st);
    for (int r = 0; r < rounds && e == hipSuccess; r++) e = hvc::launch_hd_round(P, r, st);
    if (e == hipSuccess) e = hvc::launch_hd_finish(P, rounds, st);
    return e;
}
// the `changed` word after gd_enqueue(P, rounds): the last of the launches 0 .. rounds - 1 still moved a hand-over
static bool gd_unsettled(unsigned changed_word, int rounds) { return rounds > 1 && changed_word == (unsigned)(rounds - 1); }

// ---------------------------------------------------------------------------
// Huffman decoding on the GPU (hvc_hdec.hip).  Returns HVC_OK with *used_gpu = 1 when the coefficient
// records at d_coefs are complete; HVC_OK with *used_gpu = 0 when the stream needs the host decoder
// (nothing usable was written); or the error the host decoder would report while parsing headers.
// HVC_CALL_TIMING=1 (experiments): where a single-file call spends its host time, to stderr
static bool call_timing() {
    static const bool on = std::getenv("HVC_CALL_TIMING") != nullptr;
    return on;
}
struct StageClock {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now(), last = t0;
    char line[512];
    int n = 0;
    void mark(const char *what) {
        if (!call_timing()) return;
        const auto now = std::chrono::steady_clock::now();
        n += std::snprintf(line + n, sizeof line - (size_t)n, " %s %.1f", what, std::chrono::duration<double, std::micro>(now - last).count());
        if (n > (int)sizeof line - 64) n = (int)sizeof line - 64;
        last = now;
    }
    void done() {
        if (!call_timing()) return;
        std::fprintf(stderr, "hvc call timing (us):%s | total %.1f\n", line, std::chrono::duration<double, std::micro>(last - t0).count());
    }
};

static int gpu_entropy_decode(hvc_ctx *c, const uint8_t *const *jpegs, const size_t *sizes, int n_frames,
                              const hvc_jpeg_info &info0, int16_t *d_coefs, size_t coef_fs, int *used_gpu, AfterReader *after) {
    *used_gpu = 0;
    if (after) after->speculated = false;
    StageClock clk;
    // Huffman tables per file (decoder.ml:238-259 picks them from the file's own DHT segments): the distinct sets of
    // the batch and which one every frame uses.  One set that fits two slots = the fast LDS-table kernels; anything
    // else = per-frame tables in device memory (PF mode, hvc_hdec.h).
    std::vector<hvc::HdTables> sets;
    std::vector<unsigned> tabset_of((size_t)n_frames, 0u);
    // The segments go straight from the files into ONE pinned buffer (unstuffed on the way) and from there to the
    // device: laid out by an upper bound of every segment's length -- its file's -- so that the places are known before
    // the files are read.  (Through per-file vectors, a pageable batch buffer and the runtime's own staging the bytes of
    // a 1 MB file were copied three times before the copy engine saw them: 0.15 of the call's 1.0 ms.)
    const unsigned SB = HVC_HD_SUBSEQ_BITS / 8;
    std::vector<unsigned> ecs_off((size_t)n_frames), sub_off((size_t)n_frames + 1);
    size_t bytes = 0, subs = 0;
    try {
        for (int f = 0; f < n_frames; f++) {
            const size_t nsub_most = (sizes[f] + SB - 1) / SB + 1; // an entropy-coded segment is shorter than its file
            ecs_off[(size_t)f] = (unsigned)bytes;
            bytes += nsub_most * SB + 16; // SB = 128: every frame starts on a 16-byte boundary, 16 zero bytes of overshoot
            if (bytes >= (1ull << 31)) return HVC_OK;
        }
    } catch (const std::bad_alloc &) {
        return HVC_E_OUT_OF_MEMORY;
    }
    if (bytes > c->gd_h_ecs_cap) {
        if (c->gd_h_ecs) (void)hipHostFree(c->gd_h_ecs);
        c->gd_h_ecs = nullptr;
        c->gd_h_ecs_cap = 0;
        const size_t want = bytes + bytes / 2;
        if (hipHostMalloc(&c->gd_h_ecs, want, HVC_UPLOAD_RING_FLAGS) != hipSuccess) {
            (void)hipGetLastError();
            c->gd_h_ecs = nullptr;
            return HVC_E_OUT_OF_MEMORY;
        }
        c->gd_h_ecs_cap = want;
    }
    uint8_t *const h_ecs = (uint8_t *)c->gd_h_ecs;
    try {
        hvc::HdTables t;
        for (int f = 0; f < n_frames; f++) {
            hvc_jpeg_info fi;
            int r = hvc_jpeg_read_header(jpegs[f], sizes[f], &fi);
            if (r) return r;
            if (!same_geometry(fi, info0)) return HVC_E_INVALID_ARG; // a batch shares one geometry
            bool ok = false;
            const size_t room = (sizes[f] + SB - 1) / SB * SB; // (the frame's slot without its extra subsequence and overshoot)
            size_t got = 0;
            r = hvc::prepare_gpu_decode_to(jpegs[f], sizes[f], &fi, t, h_ecs + ecs_off[(size_t)f], room, &got, ok);
            if (r) return r;
            if (!ok) return HVC_OK;
            const size_t nsub = (got + SB - 1) / SB + 1; // one extra: the reader sees zeros past the end
            std::memset(h_ecs + ecs_off[(size_t)f] + got, 0, nsub * SB + 16 - got); // (the buffer is reused from call to call)
            sub_off[(size_t)f] = (unsigned)subs;
            subs += nsub;
            if (subs >= (1ull << 31)) return HVC_OK;
            size_t k = sets.size(); // newest first: files of one source tend to come in runs
            while (k > 0 && std::memcmp(&sets[k - 1], &t, sizeof t)) k--;
            if (k == 0) {
                sets.push_back(t);
                k = sets.size();
            }
            tabset_of[(size_t)f] = (unsigned)(k - 1);
        }
    } catch (const std::bad_alloc &) {
        return HVC_E_OUT_OF_MEMORY;
    }
    clk.mark("headers+unstuff");
    const hvc::HdTables &tables0 = sets[0];
    hvc::HdParams P;
    if (!gd_geometry(info0, P)) return HVC_OK;
    P.n_frames = n_frames;
    sub_off[(size_t)n_frames] = (unsigned)subs;
    P.total_sub = (unsigned)subs;
    // the index arrays: [ecs_off n][sub_off n + 1] travel; [frame_of subs] is filled on the device from sub_off,
    // [frame_blocks n][changed, status] are written there
    const size_t meta_words = (size_t)n_frames + ((size_t)n_frames + 1) + subs + (size_t)n_frames + 2;
    std::vector<unsigned> h_meta((size_t)2 * n_frames + 1);
    for (int f = 0; f < n_frames; f++) {
        h_meta[(size_t)f] = ecs_off[(size_t)f];
        h_meta[(size_t)n_frames + (size_t)f] = sub_off[(size_t)f];
    }
    h_meta[(size_t)2 * n_frames] = sub_off[(size_t)n_frames];
    int r;
    if ((r = grow(c, &c->gd_ecs, &c->gd_ecs_cap, bytes + HVC_HD_ECS_SLACK))) return r;
    if ((r = grow(c, &c->gd_meta, &c->gd_meta_cap, meta_words * sizeof(unsigned) + 64))) return r;
    if ((r = grow(c, &c->gd_state, &c->gd_state_cap, HVC_HD_STATE_BYTES(subs)))) return r;
    if ((r = grow(c, &c->gd_dcd, &c->gd_dcd_cap, (size_t)n_frames * P.blocks_per_frame * sizeof(int16_t)))) return r;
    P.dcd = (int16_t *)c->gd_dcd;
    unsigned *m = (unsigned *)c->gd_meta;
    unsigned *d_ecs_off = m, *d_sub_off = m + n_frames, *d_frame_of = d_sub_off + n_frames + 1;
    unsigned *d_frame_blocks = d_frame_of + subs, *d_flags = d_frame_blocks + n_frames;
    hipStream_t st = c->stream;
    // From here on copies out of this function's own vectors (h_meta, ftabs, tabset_of) and out of the reused pinned
    // buffer are in flight: EVERY way out of the function waits for the stream first (the early returns included).
    std::vector<hvc::HdFrameTabs> ftabs; // (declared BEFORE the guard: destroyed after the guard has waited)
    struct SyncOnExit {
        hipStream_t s;
        ~SyncOnExit() { (void)hipStreamSynchronize(s); }
    } sync_on_exit{st};
    HIPCHK(c, hipMemcpyAsync(c->gd_ecs, h_ecs, bytes, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(m, h_meta.data(), h_meta.size() * sizeof(unsigned), hipMemcpyHostToDevice, st));
    bool pf = sets.size() > 1;
    bool any_ovf = false; // a table with prefixes beyond its sub-tables: only the fast kernels know the overflow search
    for (const hvc::HdTables &ts : sets) any_ovf |= hvc::tables_use_overflow(ts, P.n_comp);
    if (!pf) {
        if ((r = gd_upload_tables(c, tables0, P, st))) return r;
        static const bool classic = std::getenv("HVC_HD_CLASSIC") != nullptr;
        pf = !P.spec && !classic; // one set, but three different table pairs in it: no slots for that, per-component tables
        if (any_ovf && !P.spec && !pf) return HVC_OK; // (HVC_HD_CLASSIC: the general kernels -> the host reader has it)
    }
    P.coef_fs = coef_fs;
    if (any_ovf && !hvc::hd_write2_fits(P)) return HVC_OK; // (k_hd_write, the general write pass, would be chosen)
    if (pf) {
        if (!hvc::hd_write2_fits(P)) return HVC_OK; // (PF mode has the fast write pass only)
        try {
            ftabs.resize(sets.size());
        } catch (const std::bad_alloc &) {
            return HVC_E_OUT_OF_MEMORY;
        }
        for (size_t k = 0; k < sets.size(); k++) hvc::make_frame_tabs(sets[k], P.n_comp, ftabs[k]);
        const size_t tb = sets.size() * sizeof(hvc::HdFrameTabs);
        if ((r = grow(c, &c->gd_ftabs, &c->gd_ftabs_cap, tb + (size_t)n_frames * sizeof(unsigned)))) return r;
        HIPCHK(c, hipMemcpyAsync(c->gd_ftabs, ftabs.data(), tb, hipMemcpyHostToDevice, st));
        HIPCHK(c, hipMemcpyAsync((char *)c->gd_ftabs + tb, tabset_of.data(), (size_t)n_frames * sizeof(unsigned), hipMemcpyHostToDevice, st));
        P.tables = nullptr;
        P.spec = nullptr;
        P.ftabs = (const hvc::HdFrameTabs *)c->gd_ftabs;
        P.tabset_of = (const unsigned *)((char *)c->gd_ftabs + tb);
        P.selmask = gd_component_selmask(P);
        if (gd_lists_per_frame(P.total_sub, n_frames)) {
            if ((r = grow(c, &c->gd_fcnt, &c->gd_fcnt_cap, (size_t)HVC_HD_LIST_N * (size_t)n_frames * sizeof(unsigned)))) return r;
            P.list_fn = (unsigned *)c->gd_fcnt; // work lists per frame (k_hd_sync_pf)
            for (int f = 0; f < n_frames; f++) P.max_frame_sub = std::max(P.max_frame_sub, sub_off[(size_t)f + 1] - sub_off[(size_t)f]);
        }
    }
    P.ecs = (const uint8_t *)c->gd_ecs;
    P.ecs_off = d_ecs_off;
    P.sub_off = d_sub_off;
    P.frame_of = d_frame_of;
    P.coefs = d_coefs;
    P.coef_fs = coef_fs;
    gd_carve_state(P, c->gd_state, subs);
    P.frame_blocks = d_frame_blocks;
    P.changed = d_flags;
    P.status = d_flags + 1;
    if (after && after->dc_plane) {
        P.dc_plane = after->dc_plane;
        P.dc_fs = after->dc_fs;
    }
    // (no clearing of the records: the write pass stores every index of every coded block exactly once)
    // Everything in one go, as the batch pipeline does: four synchronisation launches (all of k_hd_sync's rounds count
    // as the first), the finish passes, one look at the two flags.  Only a stream that has not settled by then -- smooth
    // content can take hundreds of rounds -- is done again round by round.
    clk.mark("uploads-enqueued");
    const int first_rounds = 4;
    HIPCHK(c, gd_enqueue(P, first_rounds, st)); // (its first launch clears the flags and the list lengths)
    clk.mark("reader-enqueued");
    // The consumer of the records goes in behind the reader before anybody has looked at the reader's flags: a call
    // that waited for them first and launched the block stage afterwards stood still for 75 us in between (one file:
    // profiles/r03e_single_call_timeline_before.txt).  Records of a run that turns out unusable are garbage of the
    // right size: the consumer's output is thrown away then.
    bool consumer_enqueued = false;
    if (after && after->enqueue) {
        if ((r = after->enqueue())) return r;
        consumer_enqueued = true;
    }
    clk.mark("consumer-enqueued");
    unsigned flags[2] = {0, 0};
    HIPCHK(c, hipMemcpyAsync(flags, P.changed, sizeof flags, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    clk.mark("synchronised");
    clk.done();
#ifdef HVC_HD_STATS // experiments: entries of k_hd_sync's work lists per round; walks / inner rounds of k_hd_round's launches
    {
        unsigned ln[HVC_HD_LIST_N];
        HIPCHK(c, hipMemcpy(ln, P.list_n, sizeof ln, hipMemcpyDeviceToHost));
        std::fprintf(stderr, "hd stats: %u subsequences; lists", P.total_sub);
        for (int q = 2; q < HVC_HD_LIST_N - 2; q++) std::fprintf(stderr, " %u", ln[q]);
        std::fprintf(stderr, "; k_hd_round walks %u, inner rounds %u; changed %u status %u\n", ln[HVC_HD_LIST_N - 2], ln[HVC_HD_LIST_N - 1], flags[0], flags[1]);
        unsigned long long hs[4];
        hvc::hd_stats_read(hs);
        std::fprintf(stderr, "hd stats: round 0 walked %llu symbols (%.1f a subsequence); 64 x the longest walk of every wavefront: %llu (lanes busy %.1f %%)\n",
                     hs[0], (double)hs[0] / P.total_sub, hs[1], 100.0 * (double)hs[0] / (double)(hs[1] ? hs[1] : 1));
        const unsigned long long own = hs[2] & 0xffffffffull, over = hs[2] >> 32;
        std::fprintf(stderr, "hd stats: write pass %llu symbols inside the lanes' own subsequences + %llu beyond them (%.1f %%); 64 x trips of every wavefront: %llu (lanes busy %.1f %%)\n",
                     own, over, 100.0 * (double)over / (double)(own ? own : 1), hs[3], 100.0 * (double)(own + over) / (double)(hs[3] ? hs[3] : 1));
    }
#endif
    if (gd_unsettled(flags[0], first_rounds)) { // (the finish passes have turned the block counts into block indices: the rounds start over)
        consumer_enqueued = false; // (it ran on records the write pass never stored)
        const int max_rounds = 48;
        int round = 0;
        HIPCHK(c, hvc::launch_hd_frame_of(P, st)); // flags and list lengths cleared again
        for (;; round++) {
            HIPCHK(c, hvc::launch_hd_round(P, round, st));
            if (round >= 4) {
                unsigned changed = 0;
                HIPCHK(c, hipMemcpyAsync(&changed, P.changed, sizeof changed, hipMemcpyDeviceToHost, st));
                HIPCHK(c, hipStreamSynchronize(st));
                if (!gd_unsettled(changed, round + 1)) break;
                if (round >= max_rounds) return HVC_OK; // does not settle: let the host decoder handle it
            }
        }
        HIPCHK(c, hvc::launch_hd_finish(P, round + 1, st)); // launches 0..round have run
        HIPCHK(c, hipMemcpyAsync(flags + 1, P.status, sizeof(unsigned), hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
    }
    const unsigned status = flags[1];
    if (status) return HVC_OK; // the model raises / range / truncated stream: the host decoder reproduces it exactly
    *used_gpu = 1;
    if (after) after->speculated = consumer_enqueued;
    return HVC_OK;
}

int hvc_jpeg_entropy_decode_gpu(hvc_ctx *c, const uint8_t *const *jpegs, const size_t *sizes, int n_frames, int16_t *coefs,
                                size_t coef_fs, int where, hvc_jpeg_info *info, int *used_gpu) try {
    if (!c || !jpegs || !sizes || !coefs || !info || n_frames < 1) return HVC_E_INVALID_ARG;
    if (where != HVC_MEM_HOST && where != HVC_MEM_DEVICE) return HVC_E_INVALID_ARG;
    int r = hvc_jpeg_read_header(jpegs[0], sizes[0], info);
    if (r) return r;
    if ((n_frames > 1 && coef_fs < info->coef_count) || (coef_fs & 7)) return HVC_E_INVALID_ARG;
    DeviceGuard g(c->device);
    if (!g.ok) return fail_hip(c, hipErrorInvalidDevice);
    int16_t *d = coefs;
    const size_t total = ((size_t)(n_frames - 1) * coef_fs + info->coef_count) * sizeof(int16_t);
    if (where == HVC_MEM_HOST) {
        if ((r = grow(c, &c->gd_coefs, &c->gd_coefs_cap, total))) return r;
        d = (int16_t *)c->gd_coefs;
    } else if ((uintptr_t)coefs & 15) {
        return HVC_E_ALIGNMENT;
    }
    int gpu = 0;
    r = gpu_entropy_decode(c, jpegs, sizes, n_frames, *info, d, coef_fs, &gpu);
    if (r) return r;
    if (used_gpu) *used_gpu = gpu;
    if (gpu) {
        if (where == HVC_MEM_HOST) {
            for (int f = 0; f < n_frames; f++)
                HIPCHK(c, hipMemcpyAsync(coefs + (size_t)f * coef_fs, d + (size_t)f * coef_fs, info->coef_count * sizeof(int16_t),
                                         hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
        }
        return HVC_OK;
    }
    // host decoder (exact model behaviour for everything unusual)
    std::vector<int16_t> tmp;
    for (int f = 0; f < n_frames; f++) {
        hvc_jpeg_info fi;
        if ((r = hvc_jpeg_read_header(jpegs[f], sizes[f], &fi))) return r;
        if (fi.coef_count != info->coef_count || std::memcmp(fi.layout, info->layout, sizeof fi.layout)) return HVC_E_INVALID_ARG;
        if (where == HVC_MEM_HOST) {
            if ((r = hvc_jpeg_entropy_decode(jpegs[f], sizes[f], &fi, coefs + (size_t)f * coef_fs))) return r;
        } else {
            tmp.resize(info->coef_count);
            if ((r = hvc_jpeg_entropy_decode(jpegs[f], sizes[f], &fi, tmp.data()))) return r;
            HIPCHK(c, hipMemcpyAsync(coefs + (size_t)f * coef_fs, tmp.data(), info->coef_count * sizeof(int16_t),
                                     hipMemcpyHostToDevice, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
        }
    }
    return HVC_OK;
} HVC_ABI_CATCH

// ---------------------------------------------------------------------------
// BASELINE config 3 with the Huffman reader on the GPU as well: host threads only parse headers and
// unstuff the entropy-coded segments into a pinned ring; hipMemcpyAsync (copy stream) brings ~1 MB per
// frame to the device, where the self-synchronising decoder (hvc_hdec.hip) writes the coefficient
// records that the block stage reads.  Anything the GPU decoder hands back (unusual tables, streams the
// model treats specially, a chunk that does not settle in four launches) restarts the call on the
// host-decoder pipeline, so results and error codes are always the host decoder's.
static int decode_batch_gpu(hvc_ctx *c, const uint8_t *const *jpegs, const size_t *sizes, int n_frames, int threads,
                            int frames_per_chunk, uint8_t *pixels, size_t pixel_fs, int where, hvc_batch_stats *stats,
                            bool yuv444) {
    if (!c || !jpegs || !sizes || !pixels || n_frames < 0) return HVC_E_INVALID_ARG;
    if (where != HVC_MEM_HOST && where != HVC_MEM_DEVICE) return HVC_E_INVALID_ARG;
    if (stats) std::memset(stats, 0, sizeof *stats);
    if (n_frames == 0) return HVC_OK;
    auto host_pipeline = [&]() {
        return decode_batch_impl(c, jpegs, sizes, n_frames, threads, frames_per_chunk, pixels, pixel_fs, where, stats, yuv444);
    };
    hvc_jpeg_info info0;
    int r = hvc_jpeg_read_header(jpegs[0], sizes[0], &info0);
    if (r) return r;
    if (yuv444 && (!is_420_scan(info0) || (info0.width & 1) || (info0.height & 1))) return HVC_E_INVALID_ARG;
    const size_t out_bytes = yuv444 ? (size_t)3 * info0.width * info0.height : info0.pixel_bytes; // per frame
    if (pixel_fs < out_bytes || (!yuv444 && (pixel_fs & 7))) return HVC_E_INVALID_ARG;
    hvc::HdParams G;
    hvc::HdTables tables0;
    {
        std::vector<uint8_t> tmp;
        bool ok = false;
        r = hvc::prepare_gpu_decode(jpegs[0], sizes[0], &info0, tables0, tmp, ok);
        if (r) return r;
        if (!ok || !gd_geometry(info0, G)) return host_pipeline();
    }
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    // the reader's launches want many subsequences at once, the pipeline at least four chunks
    // (measured: 256 files best in chunks of 64, 1024 and more in chunks of 256)
    if (frames_per_chunk < 1) frames_per_chunk = n_frames / 4 < 64 ? 64 : n_frames / 4 > 256 ? 256 : n_frames / 4;
    if (frames_per_chunk > n_frames) frames_per_chunk = n_frames;
    const int C = frames_per_chunk, NB = hvc_ctx::RING;
    const int n_chunks = (n_frames + C - 1) / C;
    size_t max_file = 0;
    for (int f = 0; f < n_frames; f++) {
        if (!jpegs[f]) return HVC_E_INVALID_ARG;
        max_file = sizes[f] > max_file ? sizes[f] : max_file;
    }
    const unsigned SB = HVC_HD_SUBSEQ_BITS / 8;
    const size_t nsub_max = (max_file + SB - 1) / SB + 1;  // an entropy-coded segment is shorter than its file
    const size_t R = nsub_max * SB + 16;                   // bytes per frame in the segment ring (16-byte multiple)
    if ((size_t)C * nsub_max >= (1ull << 31) || (size_t)C * R >= (1ull << 31)) return host_pipeline();
    const size_t ecs_bytes = (size_t)C * R;
    // index arrays of a chunk: [ecs_off C][sub_off C + 1][tabset_of C][frame_of C * nsub_max][frame_blocks C][changed, status]
    const size_t meta_words = (size_t)C + ((size_t)C + 1) + (size_t)C + (size_t)C * nsub_max + (size_t)C + 2;
    const size_t meta_bytes = meta_words * sizeof(unsigned);
    const size_t ftabs_bytes = ((size_t)C + 1) * sizeof(hvc::HdFrameTabs); // record 0: the first file's tables, 1 + f: frame f's own
    const size_t coef_chunk = info0.coef_count * sizeof(int16_t) * (size_t)C;
    const size_t oring_bytes = where == HVC_MEM_HOST ? out_bytes * (size_t)C : 0;

    DeviceGuard g(c->device);
    if (!g.ok) return fail_hip(c, hipErrorInvalidDevice);
    if (!c->copy_stream) HIPCHK(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    for (int i = 0; i < NB; i++) {
        if (!c->ev_h2d[i]) HIPCHK(c, hipEventCreate(&c->ev_h2d[i]));
        if (!c->ev_kern[i]) HIPCHK(c, hipEventCreate(&c->ev_kern[i]));
    }
    for (int i = 0; i < 4; i++)
        if (!c->ev_t[i]) HIPCHK(c, hipEventCreate(&c->ev_t[i]));
    for (int i = 0; i < NB; i++)
        for (int k = 0; k < 3; k++)
            if (!c->ev_et[i][k]) HIPCHK(c, hipEventCreate(&c->ev_et[i][k])); // per-slot stage timers
    if (ecs_bytes > c->gp_ecs_bytes || meta_bytes > c->gp_meta_bytes || ftabs_bytes > c->gp_ftabs_bytes) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, hipStreamSynchronize(c->copy_stream));
        for (int i = 0; i < NB; i++) {
            if (c->gp_h_ecs[i]) (void)hipHostFree(c->gp_h_ecs[i]);
            if (c->gp_d_ecs[i]) (void)hipFree(c->gp_d_ecs[i]);
            if (c->gp_h_meta[i]) (void)hipHostFree(c->gp_h_meta[i]);
            if (c->gp_d_meta[i]) (void)hipFree(c->gp_d_meta[i]);
            if (c->gp_h_ftabs[i]) (void)hipHostFree(c->gp_h_ftabs[i]);
            if (c->gp_d_ftabs[i]) (void)hipFree(c->gp_d_ftabs[i]);
            c->gp_h_ecs[i] = c->gp_d_ecs[i] = c->gp_h_meta[i] = c->gp_d_meta[i] = c->gp_h_ftabs[i] = c->gp_d_ftabs[i] = nullptr;
        }
        c->gp_ecs_bytes = c->gp_meta_bytes = c->gp_ftabs_bytes = 0;
        for (int i = 0; i < NB; i++)
            if (hipHostMalloc(&c->gp_h_ecs[i], ecs_bytes, HVC_UPLOAD_RING_FLAGS) != hipSuccess ||
                hipMalloc(&c->gp_d_ecs[i], ecs_bytes + HVC_HD_ECS_SLACK) != hipSuccess ||
                hipHostMalloc(&c->gp_h_meta[i], meta_bytes, hipHostMallocDefault) != hipSuccess ||
                hipMalloc(&c->gp_d_meta[i], meta_bytes) != hipSuccess ||
                hipHostMalloc(&c->gp_h_ftabs[i], ftabs_bytes, HVC_UPLOAD_RING_FLAGS) != hipSuccess ||
                hipMalloc(&c->gp_d_ftabs[i], ftabs_bytes) != hipSuccess)
                return HVC_E_OUT_OF_MEMORY;
        c->gp_ecs_bytes = ecs_bytes;
        c->gp_meta_bytes = meta_bytes;
        c->gp_ftabs_bytes = ftabs_bytes;
    }
    if (coef_chunk > c->ring_bytes) { // the device coefficient ring of the host-decoder pipeline is reused
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, hipStreamSynchronize(c->copy_stream));
        for (int i = 0; i < NB; i++) {
            if (c->h_ring[i]) (void)hipHostFree(c->h_ring[i]);
            if (c->d_ring[i]) (void)hipFree(c->d_ring[i]);
            c->h_ring[i] = c->d_ring[i] = nullptr;
        }
        c->ring_bytes = 0;
        for (int i = 0; i < NB; i++)
            if (hipHostMalloc(&c->h_ring[i], coef_chunk, hipHostMallocDefault) != hipSuccess ||
                hipMalloc(&c->d_ring[i], coef_chunk) != hipSuccess)
                return HVC_E_OUT_OF_MEMORY;
        c->ring_bytes = coef_chunk;
    }
    if (oring_bytes > c->oring_bytes) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        for (int i = 0; i < NB; i++) {
            if (c->d_oring[i]) (void)hipFree(c->d_oring[i]);
            c->d_oring[i] = nullptr;
        }
        c->oring_bytes = 0;
        for (int i = 0; i < NB; i++)
            if (hipMalloc(&c->d_oring[i], oring_bytes) != hipSuccess) return HVC_E_OUT_OF_MEMORY;
        c->oring_bytes = oring_bytes;
    }
    // The reader of chunk k runs on rd_stream[k & 1] with its own per-subsequence state, the block stage of all
    // chunks on c->stream: the last synchronisation rounds of a chunk (a handful of wavefronts chasing the few
    // stretches that are slow to synchronise, each round a full kernel's latency) overlap with the next chunk's
    // first ones, which fill the GPU.
    static_assert(hvc_ctx::RING <= 3, "ev_rd");
    if (where == HVC_MEM_HOST && !c->down_stream) HIPCHK(c, hipStreamCreateWithFlags(&c->down_stream, hipStreamNonBlocking));
    if (!c->rd_stream[0]) {
        // two streams of the same priority can end up on one hardware queue (they did: no overlap at all);
        // streams of different priorities never share one
        int least = 0, greatest = 0;
        HIPCHK(c, hipDeviceGetStreamPriorityRange(&least, &greatest));
        HIPCHK(c, hipStreamCreateWithPriority(&c->rd_stream[0], hipStreamNonBlocking, least));
        HIPCHK(c, hipStreamCreateWithPriority(&c->rd_stream[1], hipStreamNonBlocking, greatest));
        HIPCHK(c, hipStreamCreateWithPriority(&c->rd_stream[2], hipStreamNonBlocking, (least + greatest) / 2));
    }
    for (int i = 0; i < NB; i++)
        if (!c->ev_rd[i]) HIPCHK(c, hipEventCreate(&c->ev_rd[i]));
#ifndef HVC_NRD
#define HVC_NRD 2
#endif
    constexpr int NRD = HVC_NRD; // reader streams in use (round 1: 1: 71 Gpixel/s on config 3, 2: 77, 3: 79 with half as much scratch again)
    const size_t state_bytes = (HVC_HD_STATE_BYTES((size_t)C * nsub_max) + 255) & ~(size_t)255;
    if ((r = grow(c, &c->gd_state, &c->gd_state_cap, NRD * state_bytes))) return r;
    if ((r = grow(c, &c->gd_fcnt, &c->gd_fcnt_cap, (size_t)NRD * HVC_HD_LIST_N * (size_t)C * sizeof(unsigned)))) return r;
    const size_t dcd_elems = ((size_t)C * G.blocks_per_frame + 127) & ~(size_t)127;
    if ((r = grow(c, &c->gd_dcd, &c->gd_dcd_cap, NRD * dcd_elems * sizeof(int16_t)))) return r;
    // The DC values go from the reader's DC pass to the block stage through a compact array, one per ring slot (a
    // chunk's block stage may still read it while the next chunk's DC pass runs), instead of 2 bytes into each
    // 128-byte record -- unless a diagnostic kernel selection asks for the A/B alternates, which read the records.
    const bool dc_compact = c->decode_kernel == 0 || c->decode_kernel == 2;
    const size_t dcv_fs = info0.coef_count / 64; // (a tight record: whole blocks)
    const size_t dcv_elems = ((size_t)C * dcv_fs + 127) & ~(size_t)127;
    if (dc_compact && (r = grow(c, &c->gd_dcv, &c->gd_dcv_cap, (size_t)NB * dcv_elems * sizeof(int16_t)))) return r;
    if ((r = gd_upload_tables(c, tables0, G, c->stream))) return r;
    // A chunk whose files all carry the first file's tables (and those fit two slots) runs on the LDS-table kernels;
    // any other chunk in PF mode (hvc_hdec.h): per-frame tables in device memory, record 0 of every ring slot = the
    // first file's, record 1 + f = frame f's own (written by the worker that unstuffs the file).
    const bool uniform_ok = G.spec != nullptr;
    const bool pf_fits = (unsigned long long)C * info0.coef_count < (1ull << 35); // hvc::hd_write2_fits for a full chunk
    if (!uniform_ok && !pf_fits) return host_pipeline();
    // tables with overflow prefixes (hvc_hdec.h HVC_HD_OVF) need the fast write pass, which a chunk this size may not fit
    if (!pf_fits && hvc::tables_use_overflow(tables0, G.n_comp)) return host_pipeline();
    for (int i = 0; i < NB; i++) hvc::make_frame_tabs(tables0, G.n_comp, *(hvc::HdFrameTabs *)c->gp_h_ftabs[i]);
    const unsigned comp_selmask = gd_component_selmask(G);
    std::vector<char> frame_pf((size_t)n_frames, 0); // the frame has tables of its own

    // workers: header parse, table check, unstuffing into the pinned segment ring
    std::mutex mu;
    std::condition_variable cv;
    std::atomic<int> next_frame{0};
    std::atomic<int> error{0};
    // chunks the GPU reader cannot or must not do (a file with other Huffman tables, tables that are no prefix code, a
    // stream the model raises on or that ends early, rounds that do not settle): skipped here or found out at the
    // verdict, and redone by the host-reader pipeline once this one has drained -- chunk by chunk, not the whole call
    std::vector<char> chunk_host((size_t)n_chunks, 0), skipped((size_t)n_chunks, 0);
    std::vector<int> done_in_chunk((size_t)n_chunks, 0);
    std::vector<unsigned> ecs_size((size_t)n_frames, 0);
    int released_upto = NB - 1;
    std::atomic<long long> prep_ns{0};
    auto worker_body = [&]() {
        hvc::HdTables t;
        if (!pin_to_ctx_cpus(c)) error.store(HVC_E_INVALID_ARG); // hvc_set_host_cpus
        for (;;) {
            const int f = next_frame.fetch_add(1);
            if (f >= n_frames || error.load()) return;
            const int k = f / C, slot = k % NB;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return k <= released_upto || error.load(); });
            }
            if (error.load()) return;
            const auto t0 = std::chrono::steady_clock::now();
            hvc_jpeg_info fi;
            int e = hvc_jpeg_read_header(jpegs[f], sizes[f], &fi);
            if (!e && (!same_geometry(fi, info0) || fi.n_qtabs != info0.n_qtabs || std::memcmp(fi.qtabs, info0.qtabs, sizeof fi.qtabs)))
                e = HVC_E_INVALID_ARG; // a batch shares one geometry and one set of quantiser tables
            bool ok = false;
            uint8_t *dst = (uint8_t *)c->gp_h_ecs[slot] + (size_t)(f - k * C) * R; // unstuffed straight into the pinned slot
            size_t got = 0;
            if (!e) e = hvc::prepare_gpu_decode_to(jpegs[f], sizes[f], &fi, t, dst, (nsub_max - 1) * SB, &got, ok);
            const bool own_tables = !e && ok && std::memcmp(&t, &tables0, sizeof t) != 0;
            const bool unfit = !e && (!ok || (own_tables && !pf_fits) || (!pf_fits && ok && hvc::tables_use_overflow(t, info0.n_comp)));
            if (own_tables && !unfit) { // its own Huffman tables: a record of its own
                hvc::make_frame_tabs(t, info0.n_comp, ((hvc::HdFrameTabs *)c->gp_h_ftabs[slot])[1 + (f - k * C)]);
                frame_pf[(size_t)f] = 1;
            }
            if (!e && !unfit) {
                const size_t used = ((got + SB - 1) / SB + 1) * SB + 16; // this frame's subsequences + overshoot
                std::memset(dst + got, 0, used - got);
                ecs_size[(size_t)f] = (unsigned)got;
            }
            prep_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
            std::lock_guard<std::mutex> lk(mu);
            if (e) error.store(e);
            if (unfit) chunk_host[(size_t)k] = 1;
            done_in_chunk[(size_t)k]++;
            cv.notify_all();
        }
    };
    auto worker = [&]() { // (a pool thread: nothing may leave it but through the error flag the orchestrator watches)
        try {
            worker_body();
        } catch (...) {
            const int e = hvc::exception_code();
            std::lock_guard<std::mutex> lk(mu);
            error.store(e);
            cv.notify_all();
        }
    };
    const auto wall0 = std::chrono::steady_clock::now();
    if ((r = pool_ready(c, threads, where == HVC_MEM_HOST ? 1 : 0))) return r;
    std::atomic<int> stage_done{0}, dl_abort{0}, dl_err{0}; // chunks whose block stage is enqueued
    std::vector<char> downloaded((size_t)n_chunks, 0);      // (everything a pool task touches is declared BEFORE the scope
                                                            // that waits for the tasks: destroyed after it has waited)
    bool completed = false; // (the workers have run out of frames, the downloader out of chunks)
    hvc::PoolScope scope(c->pool, [&] {
        std::lock_guard<std::mutex> lk(mu);
        if (!completed && !error.load()) error.store(HVC_E_INTERNAL);
        dl_abort.store(1);
        cv.notify_all();
    });

    // Host output: a thread of its own downloads chunk after chunk on c->down_stream (copies to pageable memory hold
    // their caller -- issued from the loop below they kept the next chunk's launches waiting, and on the block
    // stage's stream its kernels too: 18 Gpixel/s, 33 with this).
    auto submit_failed = [&](int e) {
        std::lock_guard<std::mutex> lk(mu);
        error.store(e);
        return e; // (the scope wakes and waits for whatever was queued)
    };
    if (where == HVC_MEM_HOST) { // (first: it must run beside the workers, never queue behind them)
        r = c->pool.submit([&] {
            (void)pin_to_ctx_cpus(c);
            if (hipSetDevice(c->device) != hipSuccess) { dl_err.store((int)hipErrorInvalidDevice); return; }
            for (int k = 0; k < n_chunks; k++) {
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return stage_done.load() > k || dl_abort.load(); });
                }
                if (dl_abort.load()) return;
                {
                    std::lock_guard<std::mutex> lk(mu);
                    if (skipped[(size_t)k]) { // nothing was decoded here: the host-reader pipeline fills it in later
                        downloaded[(size_t)k] = 1;
                        cv.notify_all();
                        continue;
                    }
                }
                const int slot = k % NB, first = k * C, cnt = (first + C <= n_frames) ? C : n_frames - first;
                hipError_t e = hipStreamWaitEvent(c->down_stream, c->ev_et[slot][2], 0);
                if (e == hipSuccess)
                    e = hipMemcpy2DAsync(pixels + (size_t)first * pixel_fs, pixel_fs, c->d_oring[slot], out_bytes, out_bytes,
                                         (size_t)cnt, hipMemcpyDeviceToHost, c->down_stream);
                if (e == hipSuccess) e = hipStreamSynchronize(c->down_stream);
                if (e != hipSuccess) dl_err.store((int)e);
                std::lock_guard<std::mutex> lk(mu);
                downloaded[(size_t)k] = 1;
                cv.notify_all();
                if (e != hipSuccess) return;
            }
        }, 1);
        if (r) return submit_failed(r);
    }
    if ((r = c->pool.submit(worker, threads))) return submit_failed(r);

    int rc = HVC_OK;
    double h2d_ms = 0, k_ms = 0;
    uint64_t ecs_total = 0;
    hipStream_t compute = c->stream;
    const bool prof_saved = c->profiling;
    c->profiling = false;
    int pending_release = -1; // the chunk whose pinned segment slot is handed on once its upload has finished
    auto release_after_upload = [&](int k) -> hipError_t {
        const hipError_t he = wait_event(c->ev_h2d[k % NB]);
        if (he != hipSuccess) return he;
        std::lock_guard<std::mutex> lk(mu);
        released_upto = k + NB;
        cv.notify_all();
        return hipSuccess;
    };
    try {
    for (int it = 0; it < n_chunks + NB && rc == HVC_OK; it++) {
        // verdict on chunk it - NB's slot before it is overwritten (and on the last chunks at the end)
        const int v = it - NB;
        if (v >= 0 && !skipped[(size_t)v]) {
            const int slot = v % NB;
            hipError_t he = hipSuccess;
            if (where == HVC_MEM_HOST) { // the slot's frames have left the device
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return downloaded[(size_t)v] != 0 || dl_err.load(); });
                if (dl_err.load()) he = (hipError_t)dl_err.load();
            } else {
                he = wait_event(c->ev_kern[slot]);
            }
            if (he != hipSuccess) { rc = fail_hip(c, he); break; }
            const unsigned *flags = (const unsigned *)c->gp_h_meta[slot] + (meta_words - 2);
            if (gd_unsettled(flags[0], 4) || flags[1]) chunk_host[(size_t)v] = 1; // not settled / the model raises / truncated: what was decoded is redone
            float ms = 0; // stage times of the chunk that just finished (read late so that nothing waits for them)
            if (hipEventElapsedTime(&ms, c->ev_et[slot][0], c->ev_h2d[slot]) == hipSuccess) h2d_ms += ms;
            if (hipEventElapsedTime(&ms, c->ev_et[slot][1], c->ev_et[slot][2]) == hipSuccess) k_ms += ms;
        }
        if (it >= n_chunks) continue;
        const int k = it, slot = k % NB, first = k * C, cnt = (first + C <= n_frames) ? C : n_frames - first;
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return done_in_chunk[(size_t)k] == cnt || error.load(); });
        }
        if (error.load()) { rc = error.load(); break; }
        bool skip;
        {
            std::lock_guard<std::mutex> lk(mu);
            skip = chunk_host[(size_t)k] != 0;
        }
        if (skip) { // no GPU work for this chunk; its pinned slot goes to chunk k + NB, the downloader moves on
            if (pending_release >= 0) { // (slots are released in order: the previous chunk's upload first)
                const hipError_t he = release_after_upload(pending_release);
                pending_release = -1;
                if (he != hipSuccess) { rc = fail_hip(c, he); break; }
            }
            std::lock_guard<std::mutex> lk(mu);
            skipped[(size_t)k] = 1;
            released_upto = k + NB;
            stage_done.store(k + 1);
            cv.notify_all();
            continue;
        }
        // the chunk's index arrays
        unsigned *hm = (unsigned *)c->gp_h_meta[slot];
        unsigned *h_ecs_off = hm, *h_sub_off = hm + C, *h_tabset_of = h_sub_off + C + 1; // (frame_of: filled on the GPU)
        unsigned subs = 0;
        bool pf = !uniform_ok;
        for (int f = 0; f < cnt; f++) {
            const unsigned nsub = (ecs_size[(size_t)(first + f)] + SB - 1) / SB + 1;
            h_tabset_of[f] = frame_pf[(size_t)(first + f)] ? 1u + (unsigned)f : 0u;
            pf |= frame_pf[(size_t)(first + f)] != 0;
            h_ecs_off[f] = (unsigned)((size_t)f * R);
            h_sub_off[f] = subs;
            subs += nsub;
            ecs_total += ecs_size[(size_t)(first + f)];
        }
        h_sub_off[cnt] = subs;
        unsigned *dm = (unsigned *)c->gp_d_meta[slot];
        hvc::HdParams P = G;
        P.n_frames = cnt;
        P.total_sub = subs;
        P.ecs = (const uint8_t *)c->gp_d_ecs[slot];
        P.ecs_off = dm;
        P.sub_off = dm + C;
        P.frame_of = dm + C + C + 1 + C;
        if (pf) {
            P.tables = nullptr;
            P.spec = nullptr;
            P.ftabs = (const hvc::HdFrameTabs *)c->gp_d_ftabs[slot];
            P.tabset_of = dm + C + C + 1;
            P.selmask = comp_selmask;
            if (gd_lists_per_frame(subs, cnt)) {
                P.list_fn = (unsigned *)c->gd_fcnt + (size_t)(k % NRD) * HVC_HD_LIST_N * (size_t)C; // work lists per frame (k_hd_sync_pf)
                P.max_frame_sub = (unsigned)nsub_max;
            }
        }
        P.frame_blocks = dm + (meta_words - 2 - C);
        P.changed = dm + (meta_words - 2);
        P.status = dm + (meta_words - 1);
        P.coefs = (int16_t *)c->d_ring[slot];
        P.coef_fs = info0.coef_count;
        gd_carve_state(P, (char *)c->gd_state + (size_t)(k % NRD) * state_bytes, (size_t)C * nsub_max);
        P.dcd = (int16_t *)c->gd_dcd + (size_t)(k % NRD) * dcd_elems;
        P.dc_plane = dc_compact ? (int16_t *)c->gd_dcv + (size_t)slot * dcv_elems : nullptr;
        P.dc_fs = dcv_fs;
        hipStream_t rs = c->rd_stream[k % NRD];
        hipError_t he = hipEventRecord(c->ev_et[slot][0], c->copy_stream);
        if (he == hipSuccess)
            he = hipMemcpyAsync(c->gp_d_ecs[slot], c->gp_h_ecs[slot], (size_t)cnt * R, hipMemcpyHostToDevice, c->copy_stream);
        if (he == hipSuccess)
            he = hipMemcpyAsync(dm, hm, ((size_t)3 * C + 1) * sizeof(unsigned), hipMemcpyHostToDevice, c->copy_stream);
        if (he == hipSuccess && pf) // the tables of the chunk's frames (36 KB a frame against ~1 MB of segment)
            he = hipMemcpyAsync(c->gp_d_ftabs[slot], c->gp_h_ftabs[slot], ((size_t)cnt + 1) * sizeof(hvc::HdFrameTabs),
                                hipMemcpyHostToDevice, c->copy_stream);
        if (he == hipSuccess) he = hipEventRecord(c->ev_h2d[slot], c->copy_stream);
        // (the slot's records and index arrays are free: the verdict above waited for chunk k - NB's block stage)
        if (he == hipSuccess) he = hipStreamWaitEvent(rs, c->ev_h2d[slot], 0);
        if (he == hipSuccess) he = hipEventRecord(c->ev_et[slot][1], rs);
        if (he == hipSuccess) he = gd_enqueue(P, 4, rs);
        if (he == hipSuccess) // changed + status -> the pinned copy of the index arrays
            he = hipMemcpyAsync(hm + (meta_words - 2), P.changed, 2 * sizeof(unsigned), hipMemcpyDeviceToHost, rs);
        if (he == hipSuccess) he = hipEventRecord(c->ev_rd[slot], rs);
        if (he == hipSuccess) he = hipStreamWaitEvent(compute, c->ev_rd[slot], 0);
        if (he != hipSuccess) { rc = fail_hip(c, he); break; }
        uint8_t *dst = where == HVC_MEM_DEVICE ? pixels + (size_t)first * pixel_fs : (uint8_t *)c->d_oring[slot];
        const size_t dst_fs = where == HVC_MEM_DEVICE ? pixel_fs : out_bytes;
        rc = yuv444 ? decode_frames_yuv444_impl(c, P.coefs, info0.coef_count, &info0.qtabs[0][0], info0.n_qtabs, info0.layout,
                                                info0.n_comp, cnt, info0.width, info0.height, dst, dst_fs, HVC_MEM_DEVICE,
                                                P.dc_plane, P.dc_fs)
                    : decode_frames_impl(c, P.coefs, info0.coef_count, &info0.qtabs[0][0], info0.n_qtabs, info0.layout,
                                         info0.n_comp, cnt, dst, dst_fs, HVC_MEM_DEVICE, P.dc_plane, P.dc_fs);
        if (rc) break;
        he = hipEventRecord(c->ev_et[slot][2], compute);
        if (he == hipSuccess && where == HVC_MEM_HOST) { // the downloader takes over
            std::lock_guard<std::mutex> lk(mu);
            stage_done.store(k + 1);
            cv.notify_all();
        }
        if (he == hipSuccess && where != HVC_MEM_HOST) he = hipEventRecord(c->ev_kern[slot], compute);
        // Hand the PREVIOUS chunk's pinned segment slot to chunk k - 1 + NB now that its upload is through -- this
        // chunk's upload is queued behind it, so the copy engine goes from one to the next while this thread waits
        // here, prepares the next chunk's index arrays and enqueues its launches (waiting for a chunk's own upload
        // at this point left the engine idle for as long as that took: 0.5 - 2 ms in every 4.8).
        if (he == hipSuccess && pending_release >= 0) he = release_after_upload(pending_release);
        if (he != hipSuccess) { rc = fail_hip(c, he); break; }
        pending_release = k;
    }
    if (rc == HVC_OK && pending_release >= 0) {
        const hipError_t he = release_after_upload(pending_release);
        if (he != hipSuccess) rc = fail_hip(c, he);
    }
    } catch (...) {
        rc = hvc::exception_code();
    }
    c->profiling = prof_saved;
    {
        std::lock_guard<std::mutex> lk(mu);
        if (rc != HVC_OK) error.store(rc);
        else completed = true;
        cv.notify_all();
    }
    { // (after a complete run the downloader has finished: the last verdicts waited for its last chunks)
        const int te = scope.finish();
        if (rc == HVC_OK && te) rc = te;
    }
    for (int i = 0; i < 3; i++) (void)hipStreamSynchronize(c->rd_stream[i]);
    (void)hipStreamSynchronize(compute);
    (void)hipStreamSynchronize(c->copy_stream);
    if (rc == HVC_OK && error.load()) rc = error.load();
    double host_entropy_ms = 0;
    for (int k = 0; k < n_chunks && rc == HVC_OK; k++) // everything has drained: the chunks left to the host reader
        if (chunk_host[(size_t)k]) {
            const int first = k * C, cnt = (first + C <= n_frames) ? C : n_frames - first;
            hvc_batch_stats hs;
            rc = decode_batch_impl(c, jpegs + first, sizes + first, cnt, threads, 0, pixels + (size_t)first * pixel_fs, pixel_fs,
                                   where, &hs, yuv444);
            host_entropy_ms += hs.entropy_ms_sum;
        }
    if (stats) {
        stats->wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
        stats->entropy_ms_sum = host_entropy_ms; // host entropy decoding: only for the chunks that fell to the host reader
        stats->host_prep_ms_sum = (double)prep_ns.load() * 1e-6;
        stats->h2d_ms_sum = h2d_ms;
        stats->kernel_ms_sum = k_ms;
        stats->chunks = n_chunks;
        stats->threads = threads;
        stats->frames_per_chunk = C;
        stats->coef_bytes = ecs_total; // bytes uploaded: the unstuffed segments
    }
    return rc;
}

int hvc_jpeg_decode_batch_gpu(hvc_ctx *c, const uint8_t *const *jpegs, const size_t *sizes, int n_frames, int threads,
                              int frames_per_chunk, uint8_t *pixels, size_t pixel_fs, int where, int yuv444,
                              hvc_batch_stats *stats) try {
    return decode_batch_gpu(c, jpegs, sizes, n_frames, threads, frames_per_chunk, pixels, pixel_fs, where, stats, yuv444 != 0);
} HVC_ABI_CATCH

// ---------------------------------------------------------------------------
// Encoder back end on the GPU: RLE + Huffman + byte stuffing of coefficient records (hvc_huff.hip)

// Geometry + scratch of one call.  `out` / `offsets` are device pointers (the caller's, or NULL = scratch
// inside ctx, see huffman_scratch_out).
static int huffman_prepare(hvc_ctx *c, const hvc_jpeg_info *info, const int16_t *d_coefs, size_t coef_fs, int n_frames,
                           uint8_t *d_out, size_t out_cap, unsigned long long *d_offsets, hvc::HuffParams &P) {
    int r = hvc_jpeg_encoder_check(info);
    if (r) return r;
    std::memset(&P, 0, sizeof P);
    const hvc_jpeg_component &c0 = info->comp[0];
    P.mbs_wide = c0.decoded_width / (8 * c0.hscale);
    P.mbs_high = c0.decoded_height / (8 * c0.vscale);
    int tile = 0, base = 0;
    for (int i = 0; i < 3; i++) {
        hvc::HuffComp &K = P.comp[i];
        K.bw = info->layout[i].blocks_w;
        K.bh = info->layout[i].blocks_h;
        K.nblk = K.bw * K.bh;
        K.tile0 = tile;
        K.h = info->comp[i].hscale;
        K.v = info->comp[i].vscale;
        K.mcu_base = base;
        K.table = info->comp[i].dc_table ? 1 : 0;
        K.coef_off = info->layout[i].coef_offset;
        tile += (K.nblk + 255) / 256;
        base += K.h * K.v;
    }
    P.tiles_per_frame = tile;
    P.blocks_per_mcu = base;
    const unsigned long long bpf = (unsigned long long)P.mbs_wide * P.mbs_high * base;
    if (bpf == 0 || bpf * 64ull * 27ull >= (1ull << 32)) return HVC_E_TOO_LARGE; // 32-bit bit offsets per frame
    P.blocks_per_frame = (unsigned)bpf;
    P.n_frames = n_frames;
    P.coefs = d_coefs;
    P.coef_fs = coef_fs;
    // worst case per block: 64 fields of 27 bits (216 bytes); the segment buffer is sized for it
    const size_t words = ((size_t)bpf * 216 + 3) / 4 + 2;
    P.bitbuf_words = (words + 15) / 16 * 16;
    P.ff_stride = P.bitbuf_words / 16;
    if (!c->hd_tables) {
        uint32_t t[2][16 + 256];
        hvc::default_enc_tables(t);
        HIPCHK(c, hipMalloc((void **)&c->hd_tables, sizeof t));
        HIPCHK(c, hipMemcpy(c->hd_tables, t, sizeof t, hipMemcpyHostToDevice));
    }
    P.tables = c->hd_tables;
    const size_t nf = (size_t)n_frames;
    if ((r = grow(c, &c->hd_lens, &c->hd_lens_cap, nf * bpf * sizeof(unsigned)))) return r;
    if ((r = grow(c, &c->hd_meta, &c->hd_meta_cap, (4 * nf + 2 * (nf + 1) + 4) * sizeof(unsigned) + 64))) return r;
    if ((r = grow(c, &c->hd_bitbuf, &c->hd_bitbuf_cap, nf * P.bitbuf_words * sizeof(unsigned)))) return r;
    if ((r = grow(c, &c->hd_ff, &c->hd_ff_cap, nf * P.ff_stride * sizeof(unsigned)))) return r;
    P.lens = (unsigned *)c->hd_lens;
    unsigned *m = (unsigned *)c->hd_meta;
    P.status = m;
    P.frame_bits = m + 4;
    P.frame_bytes = P.frame_bits + nf;
    P.frame_pieces = P.frame_bytes + nf;
    P.frame_ff = P.frame_pieces + nf;
    unsigned long long *scratch_off = (unsigned long long *)(((uintptr_t)(P.frame_ff + nf) + 7) & ~(uintptr_t)7);
    P.bitbuf = (unsigned *)c->hd_bitbuf;
    P.ff = (unsigned *)c->hd_ff;
    P.out_offsets = d_offsets ? d_offsets : scratch_off;
    P.out = d_out;
    P.out_cap = out_cap;
    return HVC_OK;
}

int hvc_jpeg_header(const hvc_jpeg_info *info, uint8_t *out, size_t cap, size_t *len) try {
    if (!info || !len || info->n_comp != 3) return HVC_E_INVALID_ARG;
    std::vector<uint8_t> o;
    hvc::jpeg_header_bytes(info, o);
    *len = o.size();
    if (!out || cap < o.size()) return HVC_E_INVALID_ARG;
    std::memcpy(out, o.data(), o.size());
    return HVC_OK;
} HVC_ABI_CATCH

int hvc_huffman_encode_frames(hvc_ctx *c, const hvc_jpeg_info *info, const int16_t *coefs, size_t coef_fs, int n_frames,
                              uint8_t *out, size_t out_cap, uint64_t *offsets, int where) try {
    if (!c || !info || !coefs || !out || !offsets || n_frames < 0 || info->n_comp != 3) return HVC_E_INVALID_ARG;
    if (where != HVC_MEM_HOST && where != HVC_MEM_DEVICE) return HVC_E_INVALID_ARG;
    if (n_frames == 0) {
        if (where == HVC_MEM_HOST) offsets[0] = 0;
        return HVC_OK;
    }
    if (n_frames > 65535) return HVC_E_TOO_LARGE;
    if (n_frames > 1 && coef_fs < info->coef_count) return HVC_E_INVALID_ARG;
    if (coef_fs & 7) return HVC_E_ALIGNMENT;
    DeviceGuard g(c->device);
    if (!g.ok) return fail_hip(c, hipErrorInvalidDevice);
    hvc::HuffParams P;
    int r;
    unsigned status = 0;
    if (where == HVC_MEM_DEVICE) {
        if (((uintptr_t)coefs & 15) || ((uintptr_t)offsets & 7)) return HVC_E_ALIGNMENT;
        if ((r = huffman_prepare(c, info, coefs, coef_fs, n_frames, out, out_cap, (unsigned long long *)offsets, P))) return r;
        HIPCHK(c, hvc::launch_huffman_encode(P, c->stream));
        HIPCHK(c, hipMemcpyAsync(&status, P.status, sizeof status, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    } else {
        const size_t cbytes = ((size_t)(n_frames - 1) * coef_fs + info->coef_count) * sizeof(int16_t);
        if ((r = grow(c, &c->d_in, &c->in_cap, cbytes))) return r;
        if ((r = grow(c, &c->hd_out, &c->hd_out_cap, out_cap))) return r;
        if ((r = huffman_prepare(c, info, (const int16_t *)c->d_in, coef_fs, n_frames, (uint8_t *)c->hd_out, out_cap, nullptr, P)))
            return r;
        HIPCHK(c, hipMemcpyAsync(c->d_in, coefs, cbytes, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hvc::launch_huffman_encode(P, c->stream));
        HIPCHK(c, hipMemcpyAsync(&status, P.status, sizeof status, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(offsets, P.out_offsets, (size_t)(n_frames + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost,
                                 c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (!(status & 7u) && offsets[n_frames] <= out_cap)
            HIPCHK(c, hipMemcpy(out, c->hd_out, (size_t)offsets[n_frames], hipMemcpyDeviceToHost));
    }
    if (status & 1u) return HVC_E_RANGE;       // a value the default tables have no code for
    if (status & 6u) return HVC_E_INVALID_ARG; // out_cap too small
    return HVC_OK;
} HVC_ABI_CATCH

// ---------------------------------------------------------------------------
// BASELINE config 5 end to end: raw frames in, JPEG files out.
//   host threads: Plane.blit_available into zero-padded planes (pinned ring)     encoder.ml:514-516
//   copy stream:  hipMemcpyAsync H2D                  compute stream: k_encode, then D2H of the coefficient records
//   host threads: write_headers + rle + write_bits + EOI per frame                encoder.ml:127-193, 371-418
// The orchestrating thread runs a three-stage software pipeline over chunks (pad k | GPU k-1 | entropy k-2).
// gpu_entropy = false: coefficient records come back to the host and host threads entropy-code them;
// gpu_entropy = true: the Huffman coder runs on the GPU too (hvc_huff.hip), only the packed segments come back
// and host threads just assemble header + segment + EOI.
static int encode_batch_impl(hvc_ctx *c, const uint8_t *const *frames, int n_frames, int width, int height, int chroma,
                             int quality, int threads, int frames_per_chunk, uint8_t *const *jpegs, const size_t *caps,
                             size_t *sizes, hvc_batch_stats *stats, bool gpu_entropy) {
    if (!c || !frames || !jpegs || !caps || !sizes || n_frames < 0) return HVC_E_INVALID_ARG;
    if (stats) std::memset(stats, 0, sizeof *stats);
    hvc_jpeg_info info;
    int r = hvc_jpeg_encoder_layout(width, height, chroma, quality, &info);
    if (r) return r;
    if ((r = hvc_jpeg_encoder_check(&info))) return r; // the model raises for this geometry
    if (n_frames == 0) return HVC_OK;
    for (int f = 0; f < n_frames; f++)
        if (!frames[f] || !jpegs[f]) return HVC_E_INVALID_ARG;
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    if (frames_per_chunk < 1) frames_per_chunk = 16;
    if (frames_per_chunk > n_frames) frames_per_chunk = n_frames;
    const int C = frames_per_chunk, NB = hvc_ctx::RING;
    const int n_chunks = (n_frames + C - 1) / C;
    const size_t pix_bytes = info.pixel_bytes, coef_bytes = info.coef_count * sizeof(int16_t);
    const size_t in_bytes = pix_bytes * (size_t)C, out_bytes = coef_bytes * (size_t)C;

    DeviceGuard g(c->device);
    if (!g.ok) return fail_hip(c, hipErrorInvalidDevice);
    if (!c->copy_stream) HIPCHK(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    if (!c->down_stream) HIPCHK(c, hipStreamCreateWithFlags(&c->down_stream, hipStreamNonBlocking));
    for (int i = 0; i < NB; i++) {
        if (!c->ev_up[i]) HIPCHK(c, hipEventCreate(&c->ev_up[i]));
        if (!c->ev_down[i]) HIPCHK(c, hipEventCreate(&c->ev_down[i]));
        for (int k = 0; k < 3; k++)
            if (!c->ev_et[i][k]) HIPCHK(c, hipEventCreate(&c->ev_et[i][k]));
        if (!c->ev_gpu[i]) HIPCHK(c, hipEventCreate(&c->ev_gpu[i]));
    }
    std::vector<uint8_t> header;
    if (gpu_entropy) {
        hvc::jpeg_header_bytes(&info, header);
        // per slot: packed segments on the device (capacity = the coefficient chunk: 2 bytes per sample, twice the raw
        // frames), and (C + 1) offsets + one status word, on the device and pinned
        const size_t off_bytes = ((size_t)C + 2) * sizeof(unsigned long long);
        if (out_bytes > c->e_seg_bytes || off_bytes > c->e_off_bytes) {
            HIPCHK(c, hipStreamSynchronize(c->stream));
            for (int i = 0; i < NB; i++) {
                if (c->ed_seg[i]) (void)hipFree(c->ed_seg[i]);
                if (c->ed_off[i]) (void)hipFree(c->ed_off[i]);
                if (c->eh_off[i]) (void)hipHostFree(c->eh_off[i]);
                c->ed_seg[i] = c->ed_off[i] = c->eh_off[i] = nullptr;
            }
            c->e_seg_bytes = c->e_off_bytes = 0;
            for (int i = 0; i < NB; i++)
                if (hipMalloc(&c->ed_seg[i], out_bytes) != hipSuccess || hipMalloc(&c->ed_off[i], off_bytes) != hipSuccess ||
                    hipHostMalloc(&c->eh_off[i], off_bytes, hipHostMallocDefault) != hipSuccess)
                    return HVC_E_OUT_OF_MEMORY;
            c->e_seg_bytes = out_bytes;
            c->e_off_bytes = off_bytes;
        }
    }
    if (in_bytes > c->e_in_bytes || out_bytes > c->e_out_bytes) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, hipStreamSynchronize(c->copy_stream));
        for (int i = 0; i < NB; i++) {
            if (c->eh_in[i]) (void)hipHostFree(c->eh_in[i]);
            if (c->eh_out[i]) (void)hipHostFree(c->eh_out[i]);
            if (c->ed_in[i]) (void)hipFree(c->ed_in[i]);
            if (c->ed_out[i]) (void)hipFree(c->ed_out[i]);
            c->eh_in[i] = c->eh_out[i] = c->ed_in[i] = c->ed_out[i] = nullptr;
        }
        c->e_in_bytes = c->e_out_bytes = 0;
        for (int i = 0; i < NB; i++)
            if (hipHostMalloc(&c->eh_in[i], in_bytes, HVC_UPLOAD_RING_FLAGS) != hipSuccess ||
                hipHostMalloc(&c->eh_out[i], out_bytes, hipHostMallocDefault) != hipSuccess ||
                hipMalloc(&c->ed_in[i], in_bytes) != hipSuccess || hipMalloc(&c->ed_out[i], out_bytes) != hipSuccess)
                return HVC_E_OUT_OF_MEMORY;
        c->e_in_bytes = in_bytes;
        c->e_out_bytes = out_bytes;
    }

    const int cw = chroma == 444 ? width : width / 2, ch = chroma == 420 ? height / 2 : height;
    const int sw[3] = {width, cw, cw}, sh[3] = {height, ch, ch};
    struct Task {
        int kind, frame; // 0 = pad into the pinned pixel ring, 1 = entropy-code from the pinned coefficient ring
    };
    std::mutex mu;
    std::condition_variable cv_task, cv_done;
    std::deque<Task> queue;
    bool stop = false;
    std::atomic<int> error{0};
    std::vector<int> pads_done((size_t)n_chunks, 0), ent_done((size_t)n_chunks, 0);
    std::atomic<long long> pad_ns{0}, ent_ns{0};
    auto worker = [&]() {
        if (!pin_to_ctx_cpus(c)) error.store(HVC_E_INVALID_ARG); // hvc_set_host_cpus
        for (;;) {
            Task t;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_task.wait(lk, [&] { return stop || !queue.empty(); });
                if (queue.empty()) return;
                t = queue.front();
                queue.pop_front();
            }
            const int f = t.frame, k = f / C, slot = k % NB;
            const auto t0 = std::chrono::steady_clock::now();
            int e = HVC_OK;
            if (error.load() == 0) try {
                if (t.kind == 0) {
                    uint8_t *rec = (uint8_t *)c->eh_in[slot] + (size_t)(f - k * C) * pix_bytes;
                    const uint8_t *src = frames[f];
                    for (int i = 0; i < 3; i++) {
                        const hvc_component &L = info.layout[i];
                        const int pw = info.comp[i].decoded_width, ph = info.comp[i].decoded_height;
                        const int bw = sw[i] < pw ? sw[i] : pw, bh = sh[i] < ph ? sh[i] : ph;
                        uint8_t *dst = rec + L.plane_offset;
                        for (int row = 0; row < ph; row++) {
                            uint8_t *d = dst + (size_t)row * L.stride;
                            if (row < bh) {
                                std::memcpy(d, src + (size_t)row * sw[i], (size_t)bw);
                                std::memset(d + bw, 0, (size_t)(pw - bw)); // Plane.create is zero-filled
                            } else {
                                std::memset(d, 0, (size_t)pw);
                            }
                        }
                        src += (size_t)sw[i] * sh[i];
                    }
                } else if (!gpu_entropy) {
                    const int16_t *cf = (const int16_t *)c->eh_out[slot] + (size_t)(f - k * C) * info.coef_count;
                    e = hvc_jpeg_entropy_encode(&info, cf, jpegs[f], caps[f], &sizes[f]);
                } else { // header + the frame's segment + EOI (complete_and_write_eoi, encoder.ml:507-510)
                    const unsigned long long *off = (const unsigned long long *)c->eh_off[slot];
                    const int fi = f - k * C;
                    const size_t seg = (size_t)(off[fi + 1] - off[fi]);
                    sizes[f] = header.size() + seg + 2;
                    if (sizes[f] > caps[f]) {
                        e = HVC_E_INVALID_ARG;
                    } else {
                        std::memcpy(jpegs[f], header.data(), header.size());
                        std::memcpy(jpegs[f] + header.size(), (const uint8_t *)c->eh_out[slot] + off[fi], seg);
                        jpegs[f][header.size() + seg] = 0xff;
                        jpegs[f][header.size() + seg + 1] = 0xd9;
                    }
                }
            } catch (...) { // (the counters below must move whatever happened: the orchestrator waits for them)
                e = hvc::exception_code();
            }
            const long long ns =
                std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
            (t.kind == 0 ? pad_ns : ent_ns) += ns;
            std::lock_guard<std::mutex> lk(mu);
            if (e) error.store(e);
            (t.kind == 0 ? pads_done : ent_done)[(size_t)k]++;
            cv_done.notify_all();
        }
    };
    const auto wall0 = std::chrono::steady_clock::now();
    if ((r = pool_ready(c, threads))) return r;
    hvc::PoolScope scope(c->pool, [&] { // however this function is left: the workers drain the queue and return
        std::lock_guard<std::mutex> lk(mu);
        stop = true;
        cv_task.notify_all();
    });
    if ((r = c->pool.submit(worker, threads))) return r;
    auto chunk_count = [&](int k) { return (k * C + C <= n_frames) ? C : n_frames - k * C; };
    auto submit = [&](int kind, int k) {
        std::lock_guard<std::mutex> lk(mu);
        for (int f = k * C; f < k * C + chunk_count(k); f++) queue.push_back(Task{kind, f});
        cv_task.notify_all();
    };
    auto wait_for = [&](std::vector<int> &done, int k) {
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [&] { return done[(size_t)k] == chunk_count(k); });
    };

    int rc = HVC_OK;
    double h2d_ms = 0, k_ms = 0, d2h_ms = 0;
    unsigned long long seg_bytes = 0;
    hipStream_t compute = c->stream;
    const bool prof_saved = c->profiling;
    c->profiling = false;
    try {
    for (int it = 0; it < n_chunks + 3 && rc == HVC_OK; it++) {
        // stage 1: pad chunk `it` (its pinned slot was uploaded and synchronised two iterations ago)
        if (it < n_chunks) submit(0, it);
        // stage 2: GPU work of chunk it - 1
        const int j = it - 1;
        if (j >= 0 && j < n_chunks) {
            const int slot = j % NB, cnt = chunk_count(j);
            wait_for(pads_done, j);
            // the slot's pinned buffers (coefficients, or offsets + segments) are free again once chunk j - NB
            // has been entropy-coded / assembled
            if (j >= NB) wait_for(ent_done, j - NB);
            hipError_t he = hipEventRecord(c->ev_et[slot][0], c->copy_stream);
            if (he == hipSuccess)
                he = hipMemcpyAsync(c->ed_in[slot], c->eh_in[slot], pix_bytes * (size_t)cnt, hipMemcpyHostToDevice,
                                    c->copy_stream);
            if (he == hipSuccess) he = hipEventRecord(c->ev_up[slot], c->copy_stream);
            if (he == hipSuccess) he = hipStreamWaitEvent(compute, c->ev_up[slot], 0);
            if (he == hipSuccess) he = hipEventRecord(c->ev_et[slot][1], compute);
            if (he != hipSuccess) { rc = fail_hip(c, he); break; }
            rc = hvc_encode_frames(c, (const uint8_t *)c->ed_in[slot], pix_bytes, &info.qtabs[0][0], info.n_qtabs,
                                   info.layout, 3, cnt, (int16_t *)c->ed_out[slot], info.coef_count, HVC_MEM_DEVICE);
            if (rc) break;
            if (!gpu_entropy) {
                he = hipEventRecord(c->ev_et[slot][2], compute);
                if (he == hipSuccess)
                    he = hipMemcpyAsync(c->eh_out[slot], c->ed_out[slot], coef_bytes * (size_t)cnt, hipMemcpyDeviceToHost,
                                        compute);
                if (he == hipSuccess) he = hipEventRecord(c->ev_down[slot], compute);
            } else {
                hvc::HuffParams HP;
                rc = huffman_prepare(c, &info, (const int16_t *)c->ed_out[slot], info.coef_count, cnt,
                                     (uint8_t *)c->ed_seg[slot], out_bytes, (unsigned long long *)c->ed_off[slot], HP);
                if (rc) break;
                he = hvc::launch_huffman_encode(HP, compute);
                if (he == hipSuccess) he = hipEventRecord(c->ev_et[slot][2], compute);
                // offsets, then the status word behind them (slot C + 1 of the pinned array)
                if (he == hipSuccess)
                    he = hipMemcpyAsync(c->eh_off[slot], c->ed_off[slot], ((size_t)cnt + 1) * sizeof(unsigned long long),
                                        hipMemcpyDeviceToHost, compute);
                if (he == hipSuccess)
                    he = hipMemcpyAsync((unsigned long long *)c->eh_off[slot] + C + 1, HP.status, sizeof(unsigned),
                                        hipMemcpyDeviceToHost, compute);
                if (he == hipSuccess) he = hipEventRecord(c->ev_gpu[slot], compute);
            }
            if (he != hipSuccess) { rc = fail_hip(c, he); break; }
        }
        // stage 3:
        //   host coder: wait for chunk it - 2's coefficients, hand them to the entropy threads
        //   GPU coder:  wait for chunk it - 2's offsets, then download exactly its packed segments -- on a
        //               stream of its own, so that the copy is not queued behind the next chunk's kernels
        const int e = it - 2;
        if (e >= 0 && e < n_chunks) {
            const int slot = e % NB;
            float ms = 0;
            if (!gpu_entropy) {
                hipError_t he = wait_event(c->ev_down[slot]);
                if (he != hipSuccess) { rc = fail_hip(c, he); break; }
                submit(1, e);
                if (hipEventElapsedTime(&ms, c->ev_et[slot][0], c->ev_up[slot]) == hipSuccess) h2d_ms += ms;
                if (hipEventElapsedTime(&ms, c->ev_et[slot][2], c->ev_down[slot]) == hipSuccess) d2h_ms += ms;
            } else {
                hipError_t he = wait_event(c->ev_gpu[slot]);
                if (he != hipSuccess) { rc = fail_hip(c, he); break; }
                if (hipEventElapsedTime(&ms, c->ev_et[slot][0], c->ev_up[slot]) == hipSuccess) h2d_ms += ms;
                const unsigned long long *off = (const unsigned long long *)c->eh_off[slot];
                const int cnt = chunk_count(e);
                const unsigned status = (unsigned)off[C + 1];
                if (status & 1u) { rc = HVC_E_RANGE; break; }              // a value without a code
                if ((status & 6u) || off[cnt] > out_bytes) { rc = HVC_E_TOO_LARGE; break; } // > 2x the raw frames
                if (e >= NB) wait_for(ent_done, e - NB); // the pinned segment slot has been assembled
                he = hipEventRecord(c->ev_et[slot][0], c->down_stream);
                if (he == hipSuccess && off[cnt])
                    he = hipMemcpyAsync(c->eh_out[slot], c->ed_seg[slot], (size_t)off[cnt], hipMemcpyDeviceToHost,
                                        c->down_stream);
                if (he == hipSuccess) he = hipEventRecord(c->ev_down[slot], c->down_stream);
                if (he != hipSuccess) { rc = fail_hip(c, he); break; }
                seg_bytes += off[cnt];
            }
            if (hipEventElapsedTime(&ms, c->ev_et[slot][1], c->ev_et[slot][2]) == hipSuccess) k_ms += ms;
        }
        // stage 4 (GPU coder): assemble the files of chunk it - 3 once its segments have landed
        const int a = it - 3;
        if (gpu_entropy && a >= 0 && a < n_chunks) {
            const int slot = a % NB;
            hipError_t he = wait_event(c->ev_down[slot]);
            if (he != hipSuccess) { rc = fail_hip(c, he); break; }
            float ms = 0;
            if (hipEventElapsedTime(&ms, c->ev_et[slot][0], c->ev_down[slot]) == hipSuccess) d2h_ms += ms;
            submit(1, a);
        }
        if (error.load()) rc = error.load();
    }
    if (rc == HVC_OK)
        for (int k = 0; k < n_chunks; k++) wait_for(ent_done, k);
    } catch (...) {
        rc = hvc::exception_code();
    }
    c->profiling = prof_saved;
    {
        std::lock_guard<std::mutex> lk(mu);
        if (rc != HVC_OK) {
            error.store(rc);
            queue.clear();
        }
        stop = true;
        cv_task.notify_all();
    }
    {
        const int te = scope.finish();
        if (rc == HVC_OK && te) rc = te;
    }
    (void)hipStreamSynchronize(compute);
    (void)hipStreamSynchronize(c->copy_stream);
    (void)hipStreamSynchronize(c->down_stream);
    if (rc == HVC_OK && error.load()) rc = error.load();
    if (stats) {
        stats->wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
        stats->entropy_ms_sum = (double)ent_ns.load() * 1e-6;
        stats->host_prep_ms_sum = (double)pad_ns.load() * 1e-6;
        stats->h2d_ms_sum = h2d_ms;
        stats->kernel_ms_sum = k_ms;
        stats->d2h_ms_sum = d2h_ms;
        stats->chunks = n_chunks;
        stats->threads = threads;
        stats->frames_per_chunk = C;
        stats->coef_bytes = gpu_entropy ? (uint64_t)seg_bytes : (uint64_t)coef_bytes * (uint64_t)n_frames; // bytes downloaded
    }
    return rc;
}

int hvc_jpeg_encode_batch(hvc_ctx *c, const uint8_t *const *frames, int n_frames, int width, int height, int chroma,
                          int quality, int threads, int frames_per_chunk, uint8_t *const *jpegs, const size_t *caps,
                          size_t *sizes, hvc_batch_stats *stats) try {
    return encode_batch_impl(c, frames, n_frames, width, height, chroma, quality, threads, frames_per_chunk, jpegs, caps,
                             sizes, stats, false);
} HVC_ABI_CATCH

int hvc_jpeg_encode_batch_gpu(hvc_ctx *c, const uint8_t *const *frames, int n_frames, int width, int height, int chroma,
                              int quality, int threads, int frames_per_chunk, uint8_t *const *jpegs, const size_t *caps,
                              size_t *sizes, hvc_batch_stats *stats) try {
    return encode_batch_impl(c, frames, n_frames, width, height, chroma, quality, threads, frames_per_chunk, jpegs, caps,
                             sizes, stats, true);
} HVC_ABI_CATCH

} // extern "C"
