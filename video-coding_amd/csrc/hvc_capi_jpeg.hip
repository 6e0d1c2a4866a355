// hvc_capi_jpeg.hip -- files through the C ABI: one at a time (hvc_jpeg_decode, hvc_jpeg_decode_yuv444, hvc_jpeg_encode) and
// BASELINE's configuration 3, the batch pipeline with the Huffman reader on the host (hvc_jpeg_decode_batch).
#include "hvc_ctx.h"

// ---------------------------------------------------------------------------
// single-frame conveniences (host memory)

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
    hvc::RestartScope honour(c->honour_restart); // (hvc_set_restart_markers; off = the model's behaviour)
    int r = hvc_jpeg_read_header(jpeg, n, info);
    if (r) return r;
    // a 4:2:0 scan: Y 2x2, Cb / Cr 1x1 (Frame.infer_chroma_subsampling, common/src/frame.ml:42-61)
    if (info->n_comp != 3 || info->comp[0].hscale != 2 || info->comp[0].vscale != 2 || info->comp[1].hscale != 1 ||
        info->comp[1].vscale != 1 || info->comp[2].hscale != 1 || info->comp[2].vscale != 1)
        return HVC_E_INVALID_ARG;
    if (frame_cap < (size_t)3 * info->width * info->height) return HVC_E_INVALID_ARG;
    if (info->width == 0 || info->height == 0) // a frame without a sample: the model decodes no block (decoder.ml:377-395)
        return hvc_jpeg_entropy_decode(jpeg, n, info, nullptr); // and of_420 of empty planes is an empty frame; the tables are still looked up
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
    if (!c || !jpeg || !info) return HVC_E_INVALID_ARG;
    hvc::RestartScope honour(c->honour_restart); // (hvc_set_restart_markers; off = the model's behaviour)
    int r = hvc_jpeg_read_header(jpeg, n, info);
    if (r) return r;
    if (pixel_cap < info->pixel_bytes || (!pixels && info->pixel_bytes)) return HVC_E_INVALID_ARG; // (planes without a sample need no memory)
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

// yuv444 = false: padded component planes per frame (hvc_jpeg_decode_batch);
// yuv444 = true: tight 4:4:4 frames through the fused kernel (hvc_jpeg_decode_batch_yuv444)
int decode_batch_impl(hvc_ctx *c, const uint8_t *const *jpegs, const size_t *sizes, int n_frames, int threads,
                             int frames_per_chunk, uint8_t *pixels, size_t pixel_fs, int where, hvc_batch_stats *stats,
                             bool yuv444) {
    if (!c || !jpegs || !sizes || !pixels || n_frames < 0) return HVC_E_INVALID_ARG;
    if (where != HVC_MEM_HOST && where != HVC_MEM_DEVICE) return HVC_E_INVALID_ARG;
    if (stats) std::memset(stats, 0, sizeof *stats);
    if (n_frames == 0) return HVC_OK;
    hvc::RestartScope honour(c->honour_restart);
    hvc_jpeg_info info0;
    int r = hvc_jpeg_read_header(jpegs[0], sizes[0], &info0);
    if (r) return r;
    if (yuv444 && (!is_420_scan(info0) || (info0.width & 1) || (info0.height & 1))) return HVC_E_INVALID_ARG;
    const size_t out_bytes = yuv444 ? (size_t)3 * info0.width * info0.height : info0.pixel_bytes; // per frame
    if (pixel_fs < out_bytes || (!yuv444 && (pixel_fs & 7))) return HVC_E_INVALID_ARG;
    if (info0.coef_count == 0) { // frames without a block (a width or height of zero): nothing to upload, nothing to launch --
        for (int f = 0; f < n_frames; f++) { // every file is still read as the model reads it (headers, tables)
            hvc_jpeg_info fi;
            if ((r = hvc_jpeg_read_header(jpegs[f], sizes[f], &fi))) return r;
            if (fi.n_comp != info0.n_comp || fi.coef_count != 0 || std::memcmp(fi.layout, info0.layout, sizeof fi.layout)) return HVC_E_INVALID_ARG;
            if ((r = hvc_jpeg_entropy_decode(jpegs[f], sizes[f], &fi, nullptr))) return r;
        }
        return HVC_OK;
    }
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
        hvc::RestartScope honour(c->honour_restart); // (a pool thread reads the files: the flag is its own)
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
