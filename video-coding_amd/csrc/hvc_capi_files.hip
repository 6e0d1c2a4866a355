// hvc_capi_files.hip -- the encoder's back end behind the C ABI: the GPU Huffman coder (hvc_huffman_encode_frames) and
// BASELINE's configuration 5 end to end, raw frames in, JPEG files out (hvc_jpeg_encode_batch, hvc_jpeg_encode_batch_gpu).
#include "hvc_ctx.h"

// ---------------------------------------------------------------------------
// Encoder back end on the GPU: RLE + Huffman + byte stuffing of coefficient records (hvc_huff.hip)

// the default code tables on the device, once per context (hvc_huff.hip reads them through HuffParams::tables)
static int upload_enc_tables(hvc_ctx *c) {
    if (c->hd_tables) return HVC_OK;
    uint32_t t[2][16 + 256];
    hvc::default_enc_tables(t);
    HIPCHK(c, hipMalloc((void **)&c->hd_tables, sizeof t));
    HIPCHK(c, hipMemcpy(c->hd_tables, t, sizeof t, hipMemcpyHostToDevice));
    return HVC_OK;
}

int hvc_huffman_code_tables(hvc_ctx *c, int table_set, int where, uint32_t *codes) try {
    if (!codes || table_set < 0 || table_set > 1) return HVC_E_INVALID_ARG;
    if (where == HVC_MEM_HOST) {
        uint32_t t[2][16 + 256];
        hvc::default_enc_tables(t);
        std::memcpy(codes, t[table_set], sizeof t[0]);
        return HVC_OK;
    }
    if (where != HVC_MEM_DEVICE || !c) return HVC_E_INVALID_ARG;
    DeviceGuard g(c->device);
    if (!g.ok) return fail_hip(c, hipErrorInvalidDevice);
    int r = upload_enc_tables(c);
    if (r) return r;
    HIPCHK(c, hipMemcpy(codes, c->hd_tables + (size_t)table_set * (16 + 256), (16 + 256) * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return HVC_OK;
} HVC_ABI_CATCH

// Geometry + scratch of one call.  `out` / `offsets` are device pointers (the caller's, or NULL = scratch
// inside ctx, see huffman_scratch_out).
int huffman_prepare(hvc_ctx *c, const hvc_jpeg_info *info, const int16_t *d_coefs, size_t coef_fs, int n_frames, uint8_t *d_out,
                    size_t out_cap, unsigned long long *d_offsets, hvc::HuffParams &P) {
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
    if ((r = upload_enc_tables(c))) return r;
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
