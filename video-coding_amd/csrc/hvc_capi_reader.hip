// hvc_capi_reader.hip -- the GPU Huffman reader (hvc_hdec.hip) behind the C ABI: hvc_jpeg_entropy_decode_gpu and the batch
// pipeline built on it (hvc_jpeg_decode_batch_gpu: host threads unstuff, the copy engine uploads, the reader and the block
// stage run per chunk).
#include "hvc_ctx.h"

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

int gpu_entropy_decode(hvc_ctx *c, const uint8_t *const *jpegs, const size_t *sizes, int n_frames, const hvc_jpeg_info &info0,
                       int16_t *d_coefs, size_t coef_fs, int *used_gpu, AfterReader *after) {
    *used_gpu = 0;
    if (after) after->speculated = false;
    StageClock clk;
    // Huffman tables per file (decoder.ml:238-259 picks them from the file's own DHT segments): the distinct sets of
    // the batch and which one every frame uses.  One set that fits two slots = the fast LDS-table kernels; anything
    // else = per-frame tables in device memory (PF mode, hvc_hdec.h).
    std::vector<hvc::HdTables> sets;
    hvc::HdParams P;
    if (!gd_geometry(info0, P)) return HVC_OK;
    // Restart intervals honoured (hvc_set_restart_markers) and the first file's scan has more than one: every interval of
    // every file is a reader frame of its own (hvc_hdec.h HdParams::rst_*) -- `units` below; else a unit is a file.
    unsigned rst = 0, ipf = 1;
    if (hvc::tl_honour_restart) {
        rst = hvc::restart_interval_of(jpegs[0], sizes[0]);
        const unsigned long long mcus = (unsigned long long)P.mbs_wide * (unsigned long long)P.mbs_high;
        if (rst && mcus > rst) {
            const unsigned long long q = (mcus + rst - 1) / rst;
            if (q * (unsigned long long)n_frames > 65535ull) return HVC_OK; // (the launches' grids: the host reader's)
            ipf = (unsigned)q;
            P.rst_mcus = rst;
            P.rst_ipf = ipf;
            P.blocks_per_frame = rst * (unsigned)P.blocks_per_mcu; // (< the file's blocks: no overflow)
        }
    }
    const size_t units = (size_t)n_frames * ipf;
    std::vector<unsigned> tabset_of;
    // The segments go straight from the files into ONE pinned buffer (unstuffed on the way) and from there to the
    // device: laid out by an upper bound of every segment's length -- its file's -- so that the places are known before
    // the files are read.  (Through per-file vectors, a pageable batch buffer and the runtime's own staging the bytes of
    // a 1 MB file were copied three times before the copy engine saw them: 0.15 of the call's 1.0 ms.)
    const unsigned SB = HVC_HD_SUBSEQ_BITS / 8;
    std::vector<unsigned> ecs_off, sub_off, file_off, uoff, ulen; // (per unit; file_off: per file)
    size_t bytes = 0, subs = 0;
    try {
        tabset_of.assign(units, 0u);
        ecs_off.resize(units);
        sub_off.resize(units + 1);
        file_off.resize((size_t)n_frames);
        uoff.resize(ipf);
        ulen.resize(ipf);
        for (int f = 0; f < n_frames; f++) {
            const size_t nsub_most = (sizes[f] + SB - 1) / SB + 1; // an entropy-coded segment is shorter than its file
            file_off[(size_t)f] = (unsigned)bytes;
            // SB = 128: every frame starts on a 16-byte boundary, 16 zero bytes of overshoot; every interval has a slot of
            // its own (hvc::hd_unit_slot: at most 2 SB + 16 bytes more than its bytes)
            bytes += ipf > 1 ? nsub_most * SB + (size_t)ipf * (2 * SB + 16) : nsub_most * SB + 16;
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
            uint8_t *const fdst = h_ecs + file_off[(size_t)f];
            if (ipf > 1) { // its intervals, each in a slot of its own (zeros behind the bytes: written with them)
                hvc::RstUnits ru{rst, ipf, uoff.data(), ulen.data()};
                r = hvc::prepare_gpu_decode_to(jpegs[f], sizes[f], &fi, t, fdst, room + (size_t)ipf * (2 * SB + 16), &got, ok, &ru);
                if (r) return r;
                if (!ok) return HVC_OK;
                for (unsigned q = 0; q < ipf; q++) {
                    const size_t u = (size_t)f * ipf + q;
                    ecs_off[u] = file_off[(size_t)f] + uoff[q];
                    sub_off[u] = (unsigned)subs;
                    subs += hvc::hd_unit_subs(ulen[q]);
                }
            } else {
                r = hvc::prepare_gpu_decode_to(jpegs[f], sizes[f], &fi, t, fdst, room, &got, ok);
                if (r) return r;
                if (!ok) return HVC_OK;
                const size_t nsub = (got + SB - 1) / SB + 1; // one extra: the reader sees zeros past the end
                std::memset(fdst + got, 0, nsub * SB + 16 - got); // (the buffer is reused from call to call)
                ecs_off[(size_t)f] = file_off[(size_t)f];
                sub_off[(size_t)f] = (unsigned)subs;
                subs += nsub;
            }
            if (subs >= (1ull << 31)) return HVC_OK;
            size_t k = sets.size(); // newest first: files of one source tend to come in runs
            while (k > 0 && std::memcmp(&sets[k - 1], &t, sizeof t)) k--;
            if (k == 0) {
                sets.push_back(t);
                k = sets.size();
            }
            for (unsigned q = 0; q < ipf; q++) tabset_of[(size_t)f * ipf + q] = (unsigned)(k - 1);
        }
    } catch (const std::bad_alloc &) {
        return HVC_E_OUT_OF_MEMORY;
    }
    clk.mark("headers+unstuff");
    const hvc::HdTables &tables0 = sets[0];
    const size_t nU = units; // the reader's frames: files, or their restart intervals
    P.n_frames = (int)nU;
    sub_off[nU] = (unsigned)subs;
    P.total_sub = (unsigned)subs;
    // the index arrays: [ecs_off n][sub_off n + 1] travel; [frame_of subs] is filled on the device from sub_off,
    // [frame_blocks n][changed, status] are written there
    const size_t meta_words = nU + (nU + 1) + subs + nU + 2;
    std::vector<unsigned> h_meta;
    try {
        h_meta.resize(2 * nU + 1);
    } catch (const std::bad_alloc &) {
        return HVC_E_OUT_OF_MEMORY;
    }
    for (size_t f = 0; f < nU; f++) {
        h_meta[f] = ecs_off[f];
        h_meta[nU + f] = sub_off[f];
    }
    h_meta[2 * nU] = sub_off[nU];
    int r;
    if ((r = grow(c, &c->gd_ecs, &c->gd_ecs_cap, bytes + HVC_HD_ECS_SLACK))) return r;
    if ((r = grow(c, &c->gd_meta, &c->gd_meta_cap, meta_words * sizeof(unsigned) + 64))) return r;
    if ((r = grow(c, &c->gd_state, &c->gd_state_cap, HVC_HD_STATE_BYTES(subs)))) return r;
    if ((r = grow(c, &c->gd_dcd, &c->gd_dcd_cap, nU * P.blocks_per_frame * sizeof(int16_t)))) return r;
    P.dcd = (int16_t *)c->gd_dcd;
    unsigned *m = (unsigned *)c->gd_meta;
    unsigned *d_ecs_off = m, *d_sub_off = m + nU, *d_frame_of = d_sub_off + nU + 1;
    unsigned *d_frame_blocks = d_frame_of + subs, *d_flags = d_frame_blocks + nU;
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
        if ((r = grow(c, &c->gd_ftabs, &c->gd_ftabs_cap, tb + nU * sizeof(unsigned)))) return r;
        HIPCHK(c, hipMemcpyAsync(c->gd_ftabs, ftabs.data(), tb, hipMemcpyHostToDevice, st));
        HIPCHK(c, hipMemcpyAsync((char *)c->gd_ftabs + tb, tabset_of.data(), nU * sizeof(unsigned), hipMemcpyHostToDevice, st));
        P.tables = nullptr;
        P.spec = nullptr;
        P.ftabs = (const hvc::HdFrameTabs *)c->gd_ftabs;
        P.tabset_of = (const unsigned *)((char *)c->gd_ftabs + tb);
        P.selmask = gd_component_selmask(P);
        if (gd_lists_per_frame(P.total_sub, n_frames)) {
            if ((r = grow(c, &c->gd_fcnt, &c->gd_fcnt_cap, (size_t)HVC_HD_LIST_N * nU * sizeof(unsigned)))) return r;
            P.list_fn = (unsigned *)c->gd_fcnt; // work lists per file (k_hd_sync_pf)
            for (size_t f = 0; f < (size_t)n_frames; f++) P.max_frame_sub = std::max(P.max_frame_sub, sub_off[(f + 1) * ipf] - sub_off[f * ipf]);
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
    static const bool hd_debug = std::getenv("HVC_HD_DEBUG") != nullptr; // (diagnostics: why a call went to the host reader)
    if (hd_debug) std::fprintf(stderr, "hd debug: frames %d subs %u changed %u status %u\n", P.n_frames, P.total_sub, flags[0], status);
    if (status) return HVC_OK; // the model raises / range / truncated stream: the host decoder reproduces it exactly
    *used_gpu = 1;
    if (after) after->speculated = consumer_enqueued;
    return HVC_OK;
}

int hvc_jpeg_entropy_decode_gpu(hvc_ctx *c, const uint8_t *const *jpegs, const size_t *sizes, int n_frames, int16_t *coefs,
                                size_t coef_fs, int where, hvc_jpeg_info *info, int *used_gpu) try {
    if (!c || !jpegs || !sizes || !coefs || !info || n_frames < 1) return HVC_E_INVALID_ARG;
    if (where != HVC_MEM_HOST && where != HVC_MEM_DEVICE) return HVC_E_INVALID_ARG;
    hvc::RestartScope honour(c->honour_restart);
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
    hvc::RestartScope honour(c->honour_restart);
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
    const unsigned SB = HVC_HD_SUBSEQ_BITS / 8;
    // Restart intervals honoured (hvc_set_restart_markers) and the first file's scan has more than one: every interval of
    // every file is a reader frame of its own (hvc_hdec.h HdParams::rst_*), ipf of them per file; else ipf = 1, frame = file.
    unsigned rst = 0, ipf = 1;
    {
        const bool geo = gd_geometry(info0, G);
        if (geo && hvc::tl_honour_restart) {
            rst = hvc::restart_interval_of(jpegs[0], sizes[0]);
            const unsigned long long mcus = (unsigned long long)G.mbs_wide * (unsigned long long)G.mbs_high;
            if (rst && mcus > rst) {
                const unsigned long long q = (mcus + rst - 1) / rst;
                if (q > 4096) return host_pipeline(); // (intervals of a few MCUs: half of every frame's subsequences would be padding)
                ipf = (unsigned)q;
                G.rst_mcus = rst;
                G.rst_ipf = ipf;
                G.blocks_per_frame = rst * (unsigned)G.blocks_per_mcu;
            }
        }
        bool ok = false;
        try {
            if (ipf > 1) {
                std::vector<uint8_t> tmp(sizes[0] + (size_t)ipf * (2 * SB + 16) + SB);
                std::vector<unsigned> uo(ipf), ul(ipf);
                hvc::RstUnits ru{rst, ipf, uo.data(), ul.data()};
                size_t got = 0;
                r = hvc::prepare_gpu_decode_to(jpegs[0], sizes[0], &info0, tables0, tmp.data(), tmp.size(), &got, ok, &ru);
            } else {
                std::vector<uint8_t> tmp;
                r = hvc::prepare_gpu_decode(jpegs[0], sizes[0], &info0, tables0, tmp, ok);
            }
        } catch (const std::bad_alloc &) {
            return HVC_E_OUT_OF_MEMORY;
        }
        if (r) return r;
        if (!ok || !geo) return host_pipeline();
    }
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    // the reader's launches want many subsequences at once, the pipeline at least four chunks
    // (measured: 256 files best in chunks of 64, 1024 and more in chunks of 256)
    if (frames_per_chunk < 1) frames_per_chunk = n_frames / 4 < 64 ? 64 : n_frames / 4 > 256 ? 256 : n_frames / 4;
    if (frames_per_chunk > n_frames) frames_per_chunk = n_frames;
    if ((unsigned long long)frames_per_chunk * ipf > 65535ull) frames_per_chunk = (int)(65535u / ipf); // (the reader's grids: frames per launch)
    const int C = frames_per_chunk, NB = hvc_ctx::RING;
    const size_t CU = (size_t)C * ipf; // reader frames of a full chunk
    const int n_chunks = (n_frames + C - 1) / C;
    size_t max_file = 0;
    for (int f = 0; f < n_frames; f++) {
        if (!jpegs[f]) return HVC_E_INVALID_ARG;
        max_file = sizes[f] > max_file ? sizes[f] : max_file;
    }
    // an entropy-coded segment is shorter than its file; every frame has one subsequence of zeros behind its bytes and 16
    // bytes of overshoot, its bytes start on a subsequence boundary (intervals: hvc::hd_unit_slot for each)
    const size_t nsub_max = ipf > 1 ? (max_file + SB - 1) / SB + 2 * (size_t)ipf : (max_file + SB - 1) / SB + 1;
    const size_t R = ipf > 1 ? (max_file + SB - 1) / SB * SB + (size_t)ipf * (2 * SB + 16) : nsub_max * SB + 16; // bytes per FILE in the segment ring (16-byte multiple)
    if ((size_t)C * nsub_max >= (1ull << 31) || (size_t)C * R >= (1ull << 31)) return host_pipeline();
    const size_t ecs_bytes = (size_t)C * R;
    // index arrays of a chunk: [ecs_off C][sub_off C + 1][tabset_of C][frame_of C * nsub_max][frame_blocks C][changed, status]
    const size_t meta_words = CU + (CU + 1) + CU + (size_t)C * nsub_max + CU + 2; // (C -> CU: per reader frame)
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
    if ((r = grow(c, &c->gd_fcnt, &c->gd_fcnt_cap, (size_t)NRD * HVC_HD_LIST_N * CU * sizeof(unsigned)))) return r;
    const size_t dcd_elems = (CU * G.blocks_per_frame + 127) & ~(size_t)127;
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
    std::vector<unsigned> unit_off, unit_len; // restart intervals: [file][interval] place inside the file's ring region, bytes
    if (ipf > 1) {
        try {
            unit_off.resize((size_t)n_frames * ipf);
            unit_len.resize((size_t)n_frames * ipf);
        } catch (const std::bad_alloc &) {
            return HVC_E_OUT_OF_MEMORY;
        }
    }
    int released_upto = NB - 1;
    std::atomic<long long> prep_ns{0};
    auto worker_body = [&]() {
        hvc::HdTables t;
        if (!pin_to_ctx_cpus(c)) error.store(HVC_E_INVALID_ARG); // hvc_set_host_cpus
        hvc::RestartScope honour(c->honour_restart);
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
            if (!e && ipf > 1) { // (every interval's slot is zero-filled behind its bytes as it is written)
                hvc::RstUnits ru{rst, ipf, unit_off.data() + (size_t)f * ipf, unit_len.data() + (size_t)f * ipf};
                e = hvc::prepare_gpu_decode_to(jpegs[f], sizes[f], &fi, t, dst, R, &got, ok, &ru);
            } else if (!e) {
                e = hvc::prepare_gpu_decode_to(jpegs[f], sizes[f], &fi, t, dst, (nsub_max - 1) * SB, &got, ok);
            }
            const bool own_tables = !e && ok && std::memcmp(&t, &tables0, sizeof t) != 0;
            const bool unfit = !e && (!ok || (own_tables && !pf_fits) || (!pf_fits && ok && hvc::tables_use_overflow(t, info0.n_comp)));
            if (own_tables && !unfit) { // its own Huffman tables: a record of its own
                hvc::make_frame_tabs(t, info0.n_comp, ((hvc::HdFrameTabs *)c->gp_h_ftabs[slot])[1 + (f - k * C)]);
                frame_pf[(size_t)f] = 1;
            }
            if (!e && !unfit) {
                if (ipf == 1) {
                    const size_t used = ((got + SB - 1) / SB + 1) * SB + 16; // this frame's subsequences + overshoot
                    std::memset(dst + got, 0, used - got);
                }
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
        unsigned *h_ecs_off = hm, *h_sub_off = hm + CU, *h_tabset_of = h_sub_off + CU + 1; // (frame_of: filled on the GPU)
        unsigned subs = 0;
        bool pf = !uniform_ok;
        for (int f = 0; f < cnt; f++) {
            pf |= frame_pf[(size_t)(first + f)] != 0;
            ecs_total += ecs_size[(size_t)(first + f)];
            for (unsigned q = 0; q < ipf; q++) { // the file's reader frames: itself, or its restart intervals
                const size_t u = (size_t)f * ipf + q, g = (size_t)(first + f) * ipf + q;
                const unsigned bytes = ipf > 1 ? unit_len[g] : ecs_size[(size_t)(first + f)];
                h_tabset_of[u] = frame_pf[(size_t)(first + f)] ? 1u + (unsigned)f : 0u;
                h_ecs_off[u] = (unsigned)((size_t)f * R + (ipf > 1 ? unit_off[g] : 0u));
                h_sub_off[u] = subs;
                subs += ipf > 1 ? (unsigned)hvc::hd_unit_subs(bytes) : (bytes + SB - 1) / SB + 1;
            }
        }
        const size_t nu = (size_t)cnt * ipf;
        h_sub_off[nu] = subs;
        unsigned *dm = (unsigned *)c->gp_d_meta[slot];
        hvc::HdParams P = G;
        P.n_frames = (int)nu;
        P.total_sub = subs;
        P.ecs = (const uint8_t *)c->gp_d_ecs[slot];
        P.ecs_off = dm;
        P.sub_off = dm + CU;
        P.frame_of = dm + CU + CU + 1 + CU;
        if (pf) {
            P.tables = nullptr;
            P.spec = nullptr;
            P.ftabs = (const hvc::HdFrameTabs *)c->gp_d_ftabs[slot];
            P.tabset_of = dm + CU + CU + 1;
            P.selmask = comp_selmask;
            if (gd_lists_per_frame(subs, cnt)) {
                P.list_fn = (unsigned *)c->gd_fcnt + (size_t)(k % NRD) * HVC_HD_LIST_N * CU; // work lists per frame (k_hd_sync_pf)
                P.max_frame_sub = (unsigned)nsub_max;
            }
        }
        P.frame_blocks = dm + (meta_words - 2 - CU);
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
            he = hipMemcpyAsync(dm, hm, (3 * CU + 1) * sizeof(unsigned), hipMemcpyHostToDevice, c->copy_stream);
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
