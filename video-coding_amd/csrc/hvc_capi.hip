// hvc_capi.hip -- the C ABI of include/hvc_jpeg.h over the gfx950 kernels: the context (streams, events, timers,
// device memory, host CPUs) and the block stage -- hvc_decode_frames, hvc_decode_frames_yuv444, hvc_encode_frames and
// their single-plane forms.  Host-side plumbing only: argument checking, geometry -> kernel parameter blocks, staging
// for host-memory callers.  The file-level entry points are in hvc_capi_jpeg.hip, hvc_capi_reader.hip, hvc_capi_files.hip;
// what they share is hvc_ctx.h.
#include "hvc_ctx.h"

bool pin_to_ctx_cpus(const hvc_ctx *c) {
    if (!c->have_cpus) { // a pool thread may still carry an earlier call's restriction
        if (c->have_default_cpus) (void)pthread_setaffinity_np(pthread_self(), sizeof(cpu_set_t), &c->default_cpus);
        return true;
    }
    return pthread_setaffinity_np(pthread_self(), sizeof(cpu_set_t), &c->cpus) == 0;
}

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

// Launches longer than about 3 ms lose 2-3 % against back-to-back shorter ones (measured on MI355X: 1080p batches of
// 2048 / 4096 frames per launch run at 71.9 / 71.6 % of the HBM peak, 1024-frame launches -- even 1900 of them back to
// back over 3 s, or sixteen of them over a 154 GB resident set -- at 74.4 %; the counters show a lower clock and more
// DRAM read-credit stalls late in a long launch, not TLB misses: DESIGN.md section 5).  So a device-memory batch is cut
// into launches of at most this many algorithmic bytes (192 B per block); HVC_LAUNCH_BYTES overrides (experiments).
size_t launch_bytes_limit() {
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
int fused444_mode() {
    static const int m = [] {
        const char *e = std::getenv("HVC_444_MODE");
        const int v = e ? std::atoi(e) : HVC_444_MODE_DEFAULT;
        return v < 0 || v > 2 ? 0 : v;
    }();
    return m;
}
// frames per launch for a batch of n_frames frames of blocks_per_frame blocks: equal parts, each within the limit
int frames_per_launch(int n_frames, unsigned long long blocks_per_frame) {
    const unsigned long long fb = blocks_per_frame * 192ull;
    unsigned long long per = fb ? launch_bytes_limit() / fb : (unsigned long long)n_frames;
    if (per < 1) per = 1;
    if (per >= (unsigned long long)n_frames) return n_frames;
    const unsigned long long parts = ((unsigned long long)n_frames + per - 1) / per;
    return (int)(((unsigned long long)n_frames + parts - 1) / parts);
}

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

// A quantiser entry of zero: the decoder multiplies by it (decoder.ml:146: the coefficient becomes 0, and the model
// decodes files whose DQT holds zeros), the encoder divides by it (encoder.ml:98-101: Division_by_zero in the model,
// HVC_E_RANGE here).
int check_qtabs(const uint16_t *qtabs, int n_qtabs, bool divides) {
    if (!qtabs || n_qtabs < 1 || n_qtabs > HVC_MAX_QTABS) return HVC_E_INVALID_ARG;
    if (divides)
        for (int i = 0; i < n_qtabs * 64; i++)
            if (qtabs[i] == 0) return HVC_E_RANGE;
    return HVC_OK;
}

hipError_t wait_event(hipEvent_t e) {
    static const bool spin = [] { const char *v = std::getenv("HVC_EVENT_SPIN"); return v && v[0] == '1'; }();
    if (spin) return hipEventSynchronize(e);
    for (int polls = 0;; polls++) {
        const hipError_t r = hipEventQuery(e);
        if (r != hipErrorNotReady) return r;
        std::this_thread::sleep_for(std::chrono::microseconds(polls < 10 ? 20 : 200));
    }
}


#ifndef HVC_KERNEL_ID
#define HVC_KERNEL_ID "unknown"
#endif
const char *hvc_version(void) { return "hvc_jpeg 0.2 (gfx950) kernels " HVC_KERNEL_ID; }

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
    case HVC_E_BUSY: return "the slot still holds a submission (hvc_wait first)";
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
    for (hvc_ctx::Slot &s : c->slots) { // (a submission nobody waited for: the streams were drained above)
        if (s.d_in) (void)hipFree(s.d_in);
        if (s.d_out) (void)hipFree(s.d_out);
        for (hipEvent_t e : {s.up0, s.up1, s.k0, s.k1, s.dn0, s.dn1})
            if (e) (void)hipEventDestroy(e);
    }
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

int hvc_set_restart_markers(hvc_ctx *c, int honour) try {
    if (!c) return HVC_E_INVALID_ARG;
    c->honour_restart = honour != 0;
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
    for (int r0 = 0; r0 < n_records; r0 += 65535) // (the grid's y dimension holds 65535 records)
        HIPCHK(c, hvc::launch_checksum(d + (size_t)r0 * record_stride, record_bytes, record_stride,
                                       n_records - r0 < 65535 ? n_records - r0 : 65535, (unsigned long long *)c->d_sums + r0, c->stream));
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

// A component without a block -- blocks_w or blocks_h of zero: what hvc_jpeg_read_header reports for a sampling factor of
// zero or a frame without width or height, the model's empty Plane.t (decoder.ml:304-345) -- has no part in the block
// stage: decode_seq never calls decode_block for it.  The decoding entry points drop such components from the list
// they work on (the others keep their offsets); *n_kept == 0: nothing to decode at all.
int drop_empty_components(const hvc_component *comps, int n_comp, hvc_component *kept, int *n_kept) {
    if (!comps || n_comp < 1 || n_comp > HVC_MAX_COMP) return HVC_E_INVALID_ARG;
    *n_kept = 0;
    for (int i = 0; i < n_comp; i++) {
        if (comps[i].blocks_w < 0 || comps[i].blocks_h < 0) return HVC_E_INVALID_ARG;
        if (comps[i].blocks_w > 0 && comps[i].blocks_h > 0) kept[(*n_kept)++] = comps[i];
    }
    return HVC_OK;
}


// ---------------------------------------------------------------------------
// Host callers get back what the kernels wrote and nothing else (hvc_ctx.h, RecordRun)
static RecordRun record_run(const hvc_component *comps, int n_comp, bool pixels) {
    int order[HVC_MAX_COMP] = {0, 1, 2, 3};
    auto at_of = [&](int i) { return pixels ? comps[i].plane_offset : comps[i].coef_offset * sizeof(int16_t); };
    for (int i = 0; i < n_comp; i++)
        for (int j = i + 1; j < n_comp; j++)
            if (at_of(order[j]) < at_of(order[i])) std::swap(order[i], order[j]);
    RecordRun r;
    size_t at = at_of(order[0]);
    r.first = at;
    for (int i = 0; i < n_comp; i++) {
        const hvc_component &k = comps[order[i]];
        if (at_of(order[i]) != at || (pixels && k.stride != (size_t)k.blocks_w * 8)) return RecordRun{};
        at += (size_t)k.blocks_w * k.blocks_h * 64 * (pixels ? 1 : sizeof(int16_t));
    }
    r.len = at - r.first;
    return r;
}
RecordRun pixel_run(const hvc_component *comps, int n_comp) { return record_run(comps, n_comp, true); }
RecordRun coef_run(const hvc_component *comps, int n_comp) { return record_run(comps, n_comp, false); }

hipError_t download_pixels(const hvc_component *comps, int n_comp, const RecordRun &run, int f0, int cnt, size_t fs,
                           const uint8_t *d_base, uint8_t *h_base, hipStream_t st) {
    if (cnt <= 0) return hipSuccess;
    if (run.len) {
        const size_t off = (size_t)f0 * fs + run.first;
        if (cnt == 1 || fs == run.len)
            return hipMemcpyAsync(h_base + off, d_base + off, (size_t)(cnt - 1) * fs + run.len, hipMemcpyDeviceToHost, st);
        return hipMemcpy2DAsync(h_base + off, fs, d_base + off, fs, run.len, (size_t)cnt, hipMemcpyDeviceToHost, st);
    }
    for (int i = 0; i < n_comp; i++) { // every plane by itself: rows of blocks_w * 8 bytes, `stride` apart, all frames' rows
        const hvc_component &k = comps[i];
        const size_t w = (size_t)k.blocks_w * 8, h = (size_t)k.blocks_h * 8;
        for (int f = f0; f < f0 + cnt; f++) {
            const size_t off = (size_t)f * fs + k.plane_offset;
            const hipError_t e = hipMemcpy2DAsync(h_base + off, k.stride, d_base + off, k.stride, w, h, hipMemcpyDeviceToHost, st);
            if (e != hipSuccess) return e;
        }
    }
    return hipSuccess;
}

hipError_t download_coefs(const hvc_component *comps, int n_comp, const RecordRun &run, int f0, int cnt, size_t fs,
                          const uint8_t *d_base, uint8_t *h_base, hipStream_t st) {
    if (cnt <= 0) return hipSuccess;
    if (run.len) {
        const size_t off = (size_t)f0 * fs + run.first;
        if (cnt == 1 || fs == run.len)
            return hipMemcpyAsync(h_base + off, d_base + off, (size_t)(cnt - 1) * fs + run.len, hipMemcpyDeviceToHost, st);
        return hipMemcpy2DAsync(h_base + off, fs, d_base + off, fs, run.len, (size_t)cnt, hipMemcpyDeviceToHost, st);
    }
    for (int f = f0; f < f0 + cnt; f++)
        for (int i = 0; i < n_comp; i++) {
            const size_t off = (size_t)f * fs + comps[i].coef_offset * sizeof(int16_t);
            const size_t n = (size_t)comps[i].blocks_w * comps[i].blocks_h * 64 * sizeof(int16_t);
            const hipError_t e = hipMemcpyAsync(h_base + off, d_base + off, n, hipMemcpyDeviceToHost, st);
            if (e != hipSuccess) return e;
        }
    return hipSuccess;
}

// A launch's grid is (tiles per frame, frames): at most 65535 frames, and the fix-up list names a block by a 32-bit id
// (frame * tiles + tile) * lanes + lane.  A batch beyond either is the caller's business no more than the 10 GB rule
// above: it is cut into launches (the largest number of frames per launch that respects all three).
static int frames_per_launch_capped(int n_frames, unsigned long long blocks_per_frame, unsigned long long ids_per_frame) {
    int per = frames_per_launch(n_frames, blocks_per_frame);
    if (per > 65535) per = 65535;
    const unsigned long long by_ids = ids_per_frame ? ((1ull << 32) - 1) / ids_per_frame : (unsigned long long)per;
    if ((unsigned long long)per > by_ids) per = (int)by_ids;
    return per < 1 ? 1 : per;
}

// dc_plane (device memory calls only, default kernels only): see hvc::DecodeParams::dc_plane
// wide (device memory calls only): blocks to recompute with their true DC once the launches are enqueued
int decode_frames_impl(hvc_ctx *c, const int16_t *coefs, size_t coef_fs, const uint16_t *qtabs, int n_qtabs,
                       const hvc_component *comps_in, int n_comp_in, int n_frames, uint8_t *pixels, size_t pixel_fs, int where,
                       const int16_t *dc_plane, size_t dc_fs, const std::vector<WideFix> *wide) {
    if (!c || n_frames < 0) return HVC_E_INVALID_ARG;
    if (where != HVC_MEM_HOST && where != HVC_MEM_DEVICE) return HVC_E_INVALID_ARG;
    int r = check_qtabs(qtabs, n_qtabs, false);
    if (r) return r;
    hvc_component kept[HVC_MAX_COMP];
    int n_comp = 0;
    if ((r = drop_empty_components(comps_in, n_comp_in, kept, &n_comp))) return r;
    if (n_comp == 0) return HVC_OK; // (no plane has a block: a record of no bytes)
    const hvc_component *const comps = kept;
    if (!coefs || !pixels) return HVC_E_INVALID_ARG;
    Layout L;
    r = make_layout(comps, n_comp, n_qtabs, L);
    if (r) return r;
    if (n_frames == 0) return HVC_OK;
    if (n_frames > 1 && (coef_fs < L.coef_span || pixel_fs < L.pixel_span)) return HVC_E_INVALID_ARG;
    if ((coef_fs & 7) || (pixel_fs & 7)) return HVC_E_ALIGNMENT;
    // frames per launch: the 10 GB rule, the grid's 65535 frames, 32-bit block ids (frames_per_launch_capped)
    const unsigned long long ids_per_frame = (unsigned long long)L.tiles_per_frame * HVC_TILE;
    const int per = frames_per_launch_capped(n_frames, L.blocks_per_frame, ids_per_frame);
    // (a side list of wide DCs names its blocks by ids of the WHOLE batch: the chunked pipelines that make one stay far below this)
    if (wide && !wide->empty() && (unsigned long long)n_frames * ids_per_frame >= (1ull << 32)) return HVC_E_TOO_LARGE;
    DeviceGuard g(c->device);
    if (!g.ok) return fail_hip(c, hipErrorInvalidDevice);

    // fix-up list: one entry per block of a LAUNCH at most (launches follow one another on the stream, each consumes its own)
    size_t need = (size_t)((unsigned long long)per * ids_per_frame);
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
    if (wide_only) c->wide_host = (long long)((unsigned long long)n_frames * L.blocks_per_frame);
    // one launch: consumes counter fix_phase, its wide kernel clears the other one (fix_assign / fix_commit above)
    auto launch = [&](hvc::DecodeParams &Q, hipEvent_t k0, hipEvent_t k1) -> hipError_t {
        if (wide_only) return hvc::launch_decode_wide_only(Q, c->stream, k0, k1);
        fix_assign(c, Q);
        const hipError_t e = hvc::launch_decode(Q, c->stream, k0, k1);
        if (e == hipSuccess) fix_commit(c);
        else fix_reset(c);
        return e;
    };
    // frames [f0, f0 + cnt) of the batch at d_coefs / d_pixels, in launches of `per` frames; the event pair brackets the
    // dominant kernel of all of them (and the few-microsecond fix-up kernels in between)
    auto launch_range = [&](const int16_t *d_coefs, uint8_t *d_pixels, int f0, int cnt, hipEvent_t k0, hipEvent_t k1) -> hipError_t {
        for (int f = f0; f < f0 + cnt; f += per) {
            hvc::DecodeParams Pk = P;
            Pk.n_frames = f0 + cnt - f < per ? f0 + cnt - f : per;
            Pk.coefs = d_coefs + (size_t)f * coef_fs;
            Pk.pixels = d_pixels + (size_t)f * pixel_fs;
            if (dc_plane) Pk.dc_plane = dc_plane + (size_t)f * dc_fs;
            const hipError_t e = launch(Pk, f == f0 ? k0 : nullptr, f + per >= f0 + cnt ? k1 : nullptr);
            if (e != hipSuccess) return e;
        }
        return hipSuccess;
    };

    if (where == HVC_MEM_DEVICE) {
        if (((uintptr_t)coefs & 15) || ((uintptr_t)pixels & 7)) return HVC_E_ALIGNMENT;
        if (dc_plane && P.kernel_sel != 0) return HVC_E_INVALID_ARG;
        P.coefs = coefs;
        P.pixels = pixels;
        P.dc_plane = dc_plane;
        P.dc_fs = dc_fs;
        const bool prof = c->profiling;
        const int slot = (int)(c->k_calls % HVC_PROF_RING);
        HIPCHK(c, launch_range(coefs, pixels, 0, n_frames, prof ? c->k0[slot] : nullptr, prof ? c->k1[slot] : nullptr));
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
    // copy back only the pixels the kernels wrote (the caller's padding stays untouched)
    const RecordRun run = pixel_run(comps, n_comp);
    if (run.len && n_frames >= 8 && cbytes >= ((size_t)64 << 20)) // large batches in the usual form: see overlapped_parts
        return overlapped_parts(
            c, n_frames,
            [&](int f0, int cnt) {
                return hipMemcpyAsync((int16_t *)c->d_in + (size_t)f0 * coef_fs, coefs + (size_t)f0 * coef_fs,
                                      ((size_t)(cnt - 1) * coef_fs + L.coef_span) * sizeof(int16_t), hipMemcpyHostToDevice, c->stream);
            },
            [&](int, int f0, int cnt) { return launch_range((const int16_t *)c->d_in, (uint8_t *)c->d_out, f0, cnt, nullptr, nullptr); },
            [&](int f0, int cnt, hipStream_t st) {
                return download_pixels(comps, n_comp, run, f0, cnt, pixel_fs, (const uint8_t *)c->d_out, pixels, st);
            });
    HIPCHK(c, hipMemcpyAsync(c->d_in, coefs, cbytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, launch_range((const int16_t *)c->d_in, (uint8_t *)c->d_out, 0, n_frames, nullptr, nullptr));
    HIPCHK(c, download_pixels(comps, n_comp, run, 0, n_frames, pixel_fs, (const uint8_t *)c->d_out, pixels, c->stream));
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
int decode_frames_yuv444_impl(hvc_ctx *c, const int16_t *coefs, size_t coef_fs, const uint16_t *qtabs, int n_qtabs,
                              const hvc_component *comps, int n_comp, int n_frames, int width, int height, uint8_t *frames,
                              size_t frame_stride, int where, const int16_t *dc_plane, size_t dc_fs,
                              const std::vector<WideFix> *wide) {
    if (!c || !coefs || !frames || !comps || n_frames < 0) return HVC_E_INVALID_ARG;
    if (where != HVC_MEM_HOST && where != HVC_MEM_DEVICE) return HVC_E_INVALID_ARG;
    int r = check_qtabs(qtabs, n_qtabs, false);
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
    // frames per launch: the 10 GB rule, the grid's 65535 frames, 32-bit block ids (frames_per_launch_capped); the fix-up
    // lists are per launch
    const unsigned long long ids_per_frame = (unsigned long long)P.tiles_per_frame * HVC_TILE * P.nw;
    const int per = frames_per_launch_capped(n_frames, L.blocks_per_frame, ids_per_frame);
    if (wide && !wide->empty() && (unsigned long long)n_frames * ids_per_frame >= (1ull << 32)) return HVC_E_TOO_LARGE;
    const unsigned long long ids = (unsigned long long)per * ids_per_frame;
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
    const size_t luma_ids = split ? (size_t)per * (size_t)((P.pl[0].cbw * P.pl[0].cbh + HVC_TILE - 1) / HVC_TILE) * HVC_TILE : 0;
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

    // frames [f0, f0 + cnt) of the batch at d_coefs / d_out, in launches of `per` frames
    auto launch_range = [&](const int16_t *d_coefs, uint8_t *d_out, int f0, int cnt, hipEvent_t k0, hipEvent_t k1) -> hipError_t {
        for (int f = f0; f < f0 + cnt; f += per) {
            hvc::Decode444Params Pk = P;
            Pk.n_frames = f0 + cnt - f < per ? f0 + cnt - f : per;
            Pk.coefs = d_coefs + (size_t)f * coef_fs;
            Pk.out = d_out + (size_t)f * frame_stride;
            if (dc_plane) Pk.dc_plane = dc_plane + (size_t)f * dc_fs;
            const hipError_t e = launch(Pk, f == f0 ? k0 : nullptr, f + per >= f0 + cnt ? k1 : nullptr);
            if (e != hipSuccess) return e;
        }
        return hipSuccess;
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
        HIPCHK(c, launch_range(coefs, frames, 0, n_frames, prof ? c->k0[slot] : nullptr, prof ? c->k1[slot] : nullptr));
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
            [&](int, int f0, int cnt) { return launch_range((const int16_t *)c->d_in, (uint8_t *)c->d_out, f0, cnt, nullptr, nullptr); },
            [&](int f0, int cnt, hipStream_t st) {
                const size_t off = (size_t)f0 * frame_stride;
                return hipMemcpy2DAsync(frames + off, frame_stride, (uint8_t *)c->d_out + off, frame_stride, out_span, (size_t)cnt,
                                        hipMemcpyDeviceToHost, st);
            });
    HIPCHK(c, hipMemcpyAsync(c->d_in, coefs, cbytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, launch_range((const int16_t *)c->d_in, (uint8_t *)c->d_out, 0, n_frames, nullptr, nullptr));
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
    // planes are "frames" of one component (a batch of any size: hvc_decode_frames cuts it into launches)
    return hvc_decode_frames(c, coefs, coef_plane_stride, qtab, 1, &comp, 1, n_planes, plane, plane_stride, where);
} HVC_ABI_CATCH

// ---------------------------------------------------------------------------
int hvc_encode_frames(hvc_ctx *c, const uint8_t *pixels, size_t pixel_fs, const uint16_t *qtabs, int n_qtabs,
                      const hvc_component *comps, int n_comp, int n_frames, int16_t *coefs, size_t coef_fs,
                      int where) try {
    if (!c || !coefs || !pixels || n_frames < 0) return HVC_E_INVALID_ARG;
    if (where != HVC_MEM_HOST && where != HVC_MEM_DEVICE) return HVC_E_INVALID_ARG;
    int r = check_qtabs(qtabs, n_qtabs, true);
    if (r) return r;
    // Encoder tables are 8-bit (Markers.Dqt element_precision = 8, encoder.ml:224-229;
    // Quant_tables.scale clips to 1..255, quant_tables.ml:139-147).
    for (int i = 0; i < n_qtabs * 64; i++)
        if (qtabs[i] > 255) return HVC_E_RANGE;
    Layout L;
    r = make_layout(comps, n_comp, n_qtabs, L);
    if (r) return r;
    if (n_frames == 0) return HVC_OK;
    if (n_frames > 1 && (coef_fs < L.coef_span || pixel_fs < L.pixel_span)) return HVC_E_INVALID_ARG;
    if ((coef_fs & 7) || (pixel_fs & 7)) return HVC_E_ALIGNMENT;
    DeviceGuard g(c->device);
    if (!g.ok) return fail_hip(c, hipErrorInvalidDevice);
    const int per = frames_per_launch_capped(n_frames, L.blocks_per_frame, 0); // the 10 GB rule and the grid's 65535 frames

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

    // frames [f0, f0 + cnt) of the batch at d_pixels / d_coefs, in launches of `per` frames
    auto launch_range = [&](const uint8_t *d_pixels, int16_t *d_coefs, int f0, int cnt, hipEvent_t k0, hipEvent_t k1) -> hipError_t {
        for (int f = f0; f < f0 + cnt; f += per) {
            hvc::EncodeParams Pk = P;
            Pk.n_frames = f0 + cnt - f < per ? f0 + cnt - f : per;
            Pk.coefs = d_coefs + (size_t)f * coef_fs;
            Pk.pixels = d_pixels + (size_t)f * pixel_fs;
            const hipError_t e = hvc::launch_encode(Pk, c->stream, f == f0 ? k0 : nullptr, f + per >= f0 + cnt ? k1 : nullptr);
            if (e != hipSuccess) return e;
        }
        return hipSuccess;
    };
    if (where == HVC_MEM_DEVICE) {
        if (((uintptr_t)coefs & 15) || ((uintptr_t)pixels & 7)) return HVC_E_ALIGNMENT;
        const bool prof = c->profiling;
        const int slot = (int)(c->k_calls % HVC_PROF_RING);
        HIPCHK(c, launch_range(pixels, coefs, 0, n_frames, prof ? c->k0[slot] : nullptr, prof ? c->k1[slot] : nullptr));
        if (prof) c->k_calls++;
        return HVC_OK;
    }

    size_t cbytes = ((size_t)(n_frames - 1) * coef_fs + L.coef_span) * sizeof(int16_t);
    size_t pbytes = (size_t)(n_frames - 1) * pixel_fs + L.pixel_span;
    r = grow(c, &c->d_in, &c->in_cap, pbytes);
    if (r) return r;
    r = grow(c, &c->d_out, &c->out_cap, cbytes);
    if (r) return r;
    // copy back only the coefficient planes (gaps in the caller's records stay untouched)
    const RecordRun run = coef_run(comps, n_comp);
    if (run.len && n_frames >= 8 && cbytes >= ((size_t)64 << 20)) // large batches in the usual form: see overlapped_parts
        return overlapped_parts(
            c, n_frames,
            [&](int f0, int cnt) {
                return hipMemcpyAsync((uint8_t *)c->d_in + (size_t)f0 * pixel_fs, pixels + (size_t)f0 * pixel_fs,
                                      (size_t)(cnt - 1) * pixel_fs + L.pixel_span, hipMemcpyHostToDevice, c->stream);
            },
            [&](int, int f0, int cnt) { return launch_range((const uint8_t *)c->d_in, (int16_t *)c->d_out, f0, cnt, nullptr, nullptr); },
            [&](int f0, int cnt, hipStream_t st) {
                return download_coefs(comps, n_comp, run, f0, cnt, coef_fs * sizeof(int16_t), (const uint8_t *)c->d_out, (uint8_t *)coefs, st);
            });
    HIPCHK(c, hipMemcpyAsync(c->d_in, pixels, pbytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, launch_range((const uint8_t *)c->d_in, (int16_t *)c->d_out, 0, n_frames, nullptr, nullptr));
    HIPCHK(c, download_coefs(comps, n_comp, run, 0, n_frames, coef_fs * sizeof(int16_t), (const uint8_t *)c->d_out, (uint8_t *)coefs, c->stream));
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
    int r = check_qtabs(qtabs, n_qtabs, true);
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
    return hvc_encode_frames(c, plane, plane_stride, qtab, 1, &comp, 1, n_planes, coefs, coef_plane_stride, where);
} HVC_ABI_CATCH

int hvc_upsample420(hvc_ctx *c, const uint8_t *src, int cw, int ch, size_t src_stride, uint8_t *dst,
                    size_t dst_stride, int n_planes, size_t src_ps, size_t dst_ps, int where) try {
    if (!c || !src || !dst || cw < 1 || ch < 1 || n_planes < 0) return HVC_E_INVALID_ARG;
    if (where != HVC_MEM_HOST && where != HVC_MEM_DEVICE) return HVC_E_INVALID_ARG;
    if (src_stride < (size_t)cw || dst_stride < (size_t)cw * 2) return HVC_E_INVALID_ARG;
    if (n_planes == 0) return HVC_OK;
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
    auto launch_all = [&](const uint8_t *d_src, uint8_t *d_dst) -> hipError_t { // (the grid's y dimension holds 65535 planes)
        for (int p0 = 0; p0 < n_planes; p0 += 65535) {
            hvc::UpsampleParams Q = P;
            Q.n_planes = n_planes - p0 < 65535 ? n_planes - p0 : 65535;
            Q.src = d_src + (size_t)p0 * src_ps;
            Q.dst = d_dst + (size_t)p0 * dst_ps;
            const hipError_t e = hvc::launch_upsample420(Q, c->stream);
            if (e != hipSuccess) return e;
        }
        return hipSuccess;
    };
    if (where == HVC_MEM_DEVICE) {
        HIPCHK(c, launch_all(src, dst));
        return HVC_OK;
    }
    size_t sbytes = (size_t)(n_planes - 1) * src_ps + (size_t)(ch - 1) * src_stride + (size_t)cw;
    size_t dbytes = (size_t)(n_planes - 1) * dst_ps + (size_t)(2 * ch - 1) * dst_stride + (size_t)cw * 2;
    int r = grow(c, &c->d_in, &c->in_cap, sbytes);
    if (r) return r;
    r = grow(c, &c->d_out, &c->out_cap, dbytes);
    if (r) return r;
    HIPCHK(c, hipMemcpyAsync(c->d_in, src, sbytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, launch_all((const uint8_t *)c->d_in, (uint8_t *)c->d_out));
    for (int p = 0; p < n_planes; p++)
        HIPCHK(c, hipMemcpy2DAsync(dst + (size_t)p * dst_ps, dst_stride, (uint8_t *)c->d_out + (size_t)p * dst_ps,
                                   dst_stride, (size_t)cw * 2, (size_t)ch * 2, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return HVC_OK;
} HVC_ABI_CATCH
