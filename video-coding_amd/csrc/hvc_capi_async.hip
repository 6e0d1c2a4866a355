// hvc_capi_async.hip -- the asynchronous seam of include/hvc_jpeg.h: pinned host memory and the slots behind
// hvc_decode_frames_submit / hvc_encode_frames_submit / hvc_wait (SURVEY.md 8b's "submit(frame batch, stream slot) /
// wait(slot)"; BASELINE.json north_star's "pinned coefficient buffers via hipMemcpyAsync on a side stream overlapped with the
// IDCT kernel").  For the caller that keeps its own Huffman reader -- the model's, decoder.ml:118-140 -- and fills the next
// batch while the GPU works on this one.  Plumbing only: the arithmetic is hvc_decode_frames' / hvc_encode_frames'
// (hvc_capi.hip), called on the slot's device buffers.
//
//   copy_stream:   slot s upload   ──ev up1──┐
//   c->stream:                               └─> block stage of s ──ev k1──┐              (kernels of all slots in submission order)
//   down_stream:                                                           └─> download of s ──ev dn1 = the slot is done
//
// A slot's device buffers belong to it alone, so slot s + 1's upload runs under slot s's kernels and slot s - 1's
// download: both directions of the link and the kernel overlap.
#include "hvc_ctx.h"

#include <unistd.h>

namespace {

int slot_of(hvc_ctx *c, int slot, hvc_ctx::Slot **out) {
    if (!c || slot < 0 || slot >= HVC_SLOTS) return HVC_E_INVALID_ARG;
    *out = &c->slots[slot];
    return HVC_OK;
}

// streams and the slot's events, created by the first submission that needs them
int slot_prepare(hvc_ctx *c, hvc_ctx::Slot &s) {
    if (!c->copy_stream) HIPCHK(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    if (!c->down_stream) HIPCHK(c, hipStreamCreateWithFlags(&c->down_stream, hipStreamNonBlocking));
    hipEvent_t *const ev[6] = {&s.up0, &s.up1, &s.k0, &s.k1, &s.dn0, &s.dn1};
    for (hipEvent_t *e : ev)
        if (!*e) HIPCHK(c, hipEventCreate(e));
    return HVC_OK;
}

// the slot is free: nothing of it is in flight, its buffers may be replaced
int slot_grow(hvc_ctx *c, void **p, size_t *cap, size_t need) {
    if (need <= *cap) return HVC_OK;
    if (*p) {
        HIPCHK(c, hipFree(*p));
        *p = nullptr;
        *cap = 0;
    }
    const size_t want = need + need / 8 + 4096;
    const hipError_t e = hipMalloc(p, want);
    if (e != hipSuccess) {
        *p = nullptr;
        c->last_hip = (int)e;
        return HVC_E_OUT_OF_MEMORY;
    }
    *cap = want;
    return HVC_OK;
}

// Something of a submission may have reached the streams when a later step failed: drain them, so that the slot (left
// free) holds nothing in flight and the caller's buffers are its own again.
int abandon(hvc_ctx *c, int code) {
    (void)hipStreamSynchronize(c->copy_stream);
    (void)hipStreamSynchronize(c->stream);
    (void)hipStreamSynchronize(c->down_stream);
    return code;
}

} // namespace

int hvc_host_alloc(hvc_ctx *c, size_t bytes, void **out) try {
    if (!c || !out) return HVC_E_INVALID_ARG;
    *out = nullptr;
    DeviceGuard g(c->device);
    if (!g.ok) return fail_hip(c, hipErrorInvalidDevice);
    const hipError_t e = hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault);
    if (e != hipSuccess) {
        *out = nullptr;
        c->last_hip = (int)e;
        return HVC_E_OUT_OF_MEMORY;
    }
    return HVC_OK;
} HVC_ABI_CATCH

int hvc_host_free(hvc_ctx *c, void *p) try {
    if (!c) return HVC_E_INVALID_ARG;
    if (!p) return HVC_OK;
    DeviceGuard g(c->device);
    HIPCHK(c, hipHostFree(p));
    return HVC_OK;
} HVC_ABI_CATCH

// Whole pages only.  The runtime resolves a host pointer to a registered range by PAGE: a pageable buffer that merely
// starts in the last page of somebody's registered range is taken for part of it, copied through that range's mapping, and
// faults on the GPU where the mapping ends (found by tools/stress_seam.py with registered and pageable arrays side by side
// on the heap: the fault address was the first page behind a registered array, 3.6 KB past its end).  A range of whole
// pages that the caller owns outright cannot be shared that way.
int hvc_host_register(hvc_ctx *c, void *p, size_t bytes) try {
    if (!c || !p || !bytes) return HVC_E_INVALID_ARG;
    const size_t page = (size_t)sysconf(_SC_PAGESIZE);
    if (((uintptr_t)p & (page - 1)) || (bytes & (page - 1))) return HVC_E_ALIGNMENT;
    DeviceGuard g(c->device);
    if (!g.ok) return fail_hip(c, hipErrorInvalidDevice);
    HIPCHK(c, hipHostRegister(p, bytes, hipHostRegisterDefault));
    return HVC_OK;
} HVC_ABI_CATCH

int hvc_host_unregister(hvc_ctx *c, void *p) try {
    if (!c || !p) return HVC_E_INVALID_ARG;
    DeviceGuard g(c->device);
    HIPCHK(c, hipHostUnregister(p));
    return HVC_OK;
} HVC_ABI_CATCH

int hvc_decode_frames_submit(hvc_ctx *c, int slot, const int16_t *coefs, size_t coef_fs, const uint16_t *qtabs, int n_qtabs,
                             const hvc_component *comps_in, int n_comp_in, int n_frames, uint8_t *pixels, size_t pixel_fs,
                             int pixels_where) try {
    hvc_ctx::Slot *sp = nullptr;
    int r = slot_of(c, slot, &sp);
    if (r) return r;
    hvc_ctx::Slot &s = *sp;
    if (s.busy) return HVC_E_BUSY;
    if (n_frames < 0 || (pixels_where != HVC_MEM_HOST && pixels_where != HVC_MEM_DEVICE)) return HVC_E_INVALID_ARG;
    // the checks of hvc_decode_frames that decide how many bytes travel (everything else it repeats itself)
    if ((r = check_qtabs(qtabs, n_qtabs, false))) return r;
    hvc_component kept[HVC_MAX_COMP];
    int n_comp = 0;
    if ((r = drop_empty_components(comps_in, n_comp_in, kept, &n_comp))) return r;
    if (n_comp == 0 || n_frames == 0) return HVC_OK; // nothing to decode: the slot stays free
    if (!coefs || !pixels) return HVC_E_INVALID_ARG;
    Layout L;
    if ((r = make_layout(kept, n_comp, n_qtabs, L))) return r;
    if (n_frames > 1 && (coef_fs < L.coef_span || pixel_fs < L.pixel_span)) return HVC_E_INVALID_ARG;
    if ((coef_fs & 7) || (pixel_fs & 7)) return HVC_E_ALIGNMENT;
    if (pixels_where == HVC_MEM_DEVICE && ((uintptr_t)pixels & 7)) return HVC_E_ALIGNMENT;
    DeviceGuard g(c->device);
    if (!g.ok) return fail_hip(c, hipErrorInvalidDevice);
    if ((r = slot_prepare(c, s))) return r;

    const size_t cbytes = ((size_t)(n_frames - 1) * coef_fs + L.coef_span) * sizeof(int16_t);
    const size_t pbytes = (size_t)(n_frames - 1) * pixel_fs + L.pixel_span;
    const bool down = pixels_where == HVC_MEM_HOST;
    if ((r = slot_grow(c, &s.d_in, &s.in_cap, cbytes))) return r;
    if (down && (r = slot_grow(c, &s.d_out, &s.out_cap, pbytes))) return r;
    uint8_t *const d_pixels = down ? (uint8_t *)s.d_out : pixels;

    // upload on the copy stream; the block stage waits for it on the context's stream
    HIPCHK(c, hipEventRecord(s.up0, c->copy_stream));
    hipError_t e = hipMemcpyAsync(s.d_in, coefs, cbytes, hipMemcpyHostToDevice, c->copy_stream);
    if (e == hipSuccess) e = hipEventRecord(s.up1, c->copy_stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(c->stream, s.up1, 0);
    if (e == hipSuccess) e = hipEventRecord(s.k0, c->stream);
    if (e != hipSuccess) return abandon(c, fail_hip(c, e));
    // (the device-memory form of hvc_decode_frames: launches enqueued on c->stream, nothing waited for; its own event ring
    // is for hvc_set_profiling's callers, the slot has its pair)
    const bool prof_saved = c->profiling;
    c->profiling = false;
    r = decode_frames_impl(c, (const int16_t *)s.d_in, coef_fs, qtabs, n_qtabs, kept, n_comp, n_frames, d_pixels, pixel_fs,
                           HVC_MEM_DEVICE, nullptr, 0);
    c->profiling = prof_saved;
    if (r) return abandon(c, r);
    e = hipEventRecord(s.k1, c->stream);
    if (e == hipSuccess && down) {
        e = hipStreamWaitEvent(c->down_stream, s.k1, 0);
        if (e == hipSuccess) e = hipEventRecord(s.dn0, c->down_stream);
        if (e == hipSuccess)
            e = download_pixels(kept, n_comp, pixel_run(kept, n_comp), 0, n_frames, pixel_fs, (const uint8_t *)s.d_out, pixels,
                                c->down_stream);
        if (e == hipSuccess) e = hipEventRecord(s.dn1, c->down_stream);
    }
    if (e != hipSuccess) return abandon(c, fail_hip(c, e));
    s.busy = true;
    s.has_down = down;
    s.timed = false;
    s.h2d_bytes = cbytes;
    s.d2h_bytes = down ? (unsigned long long)n_frames * L.blocks_per_frame * 64 : 0;
    return HVC_OK;
} HVC_ABI_CATCH

int hvc_encode_frames_submit(hvc_ctx *c, int slot, const uint8_t *pixels, size_t pixel_fs, const uint16_t *qtabs, int n_qtabs,
                             const hvc_component *comps, int n_comp, int n_frames, int16_t *coefs, size_t coef_fs,
                             int coefs_where) try {
    hvc_ctx::Slot *sp = nullptr;
    int r = slot_of(c, slot, &sp);
    if (r) return r;
    hvc_ctx::Slot &s = *sp;
    if (s.busy) return HVC_E_BUSY;
    if (!coefs || !pixels || n_frames < 0) return HVC_E_INVALID_ARG;
    if (coefs_where != HVC_MEM_HOST && coefs_where != HVC_MEM_DEVICE) return HVC_E_INVALID_ARG;
    if ((r = check_qtabs(qtabs, n_qtabs, true))) return r;
    for (int i = 0; i < n_qtabs * 64; i++)
        if (qtabs[i] > 255) return HVC_E_RANGE; // (as hvc_encode_frames: the encoder's tables are 8-bit)
    Layout L;
    if ((r = make_layout(comps, n_comp, n_qtabs, L))) return r;
    if (n_frames == 0) return HVC_OK;
    if (n_frames > 1 && (coef_fs < L.coef_span || pixel_fs < L.pixel_span)) return HVC_E_INVALID_ARG;
    if ((coef_fs & 7) || (pixel_fs & 7)) return HVC_E_ALIGNMENT;
    if (coefs_where == HVC_MEM_DEVICE && ((uintptr_t)coefs & 15)) return HVC_E_ALIGNMENT;
    DeviceGuard g(c->device);
    if (!g.ok) return fail_hip(c, hipErrorInvalidDevice);
    if ((r = slot_prepare(c, s))) return r;

    const size_t cbytes = ((size_t)(n_frames - 1) * coef_fs + L.coef_span) * sizeof(int16_t);
    const size_t pbytes = (size_t)(n_frames - 1) * pixel_fs + L.pixel_span;
    const bool down = coefs_where == HVC_MEM_HOST;
    if ((r = slot_grow(c, &s.d_in, &s.in_cap, pbytes))) return r;
    if (down && (r = slot_grow(c, &s.d_out, &s.out_cap, cbytes))) return r;
    int16_t *const d_coefs = down ? (int16_t *)s.d_out : coefs;

    HIPCHK(c, hipEventRecord(s.up0, c->copy_stream));
    hipError_t e = hipMemcpyAsync(s.d_in, pixels, pbytes, hipMemcpyHostToDevice, c->copy_stream);
    if (e == hipSuccess) e = hipEventRecord(s.up1, c->copy_stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(c->stream, s.up1, 0);
    if (e == hipSuccess) e = hipEventRecord(s.k0, c->stream);
    if (e != hipSuccess) return abandon(c, fail_hip(c, e));
    const bool prof_saved = c->profiling;
    c->profiling = false;
    r = hvc_encode_frames(c, (const uint8_t *)s.d_in, pixel_fs, qtabs, n_qtabs, comps, n_comp, n_frames, d_coefs, coef_fs, HVC_MEM_DEVICE);
    c->profiling = prof_saved;
    if (r) return abandon(c, r);
    e = hipEventRecord(s.k1, c->stream);
    if (e == hipSuccess && down) {
        e = hipStreamWaitEvent(c->down_stream, s.k1, 0);
        if (e == hipSuccess) e = hipEventRecord(s.dn0, c->down_stream);
        if (e == hipSuccess)
            e = download_coefs(comps, n_comp, coef_run(comps, n_comp), 0, n_frames, coef_fs * sizeof(int16_t), (const uint8_t *)s.d_out,
                               (uint8_t *)coefs, c->down_stream);
        if (e == hipSuccess) e = hipEventRecord(s.dn1, c->down_stream);
    }
    if (e != hipSuccess) return abandon(c, fail_hip(c, e));
    s.busy = true;
    s.has_down = down;
    s.timed = false;
    s.h2d_bytes = pbytes;
    s.d2h_bytes = down ? (unsigned long long)n_frames * L.blocks_per_frame * 128 : 0;
    return HVC_OK;
} HVC_ABI_CATCH

int hvc_wait(hvc_ctx *c, int slot) try {
    hvc_ctx::Slot *sp = nullptr;
    int r = slot_of(c, slot, &sp);
    if (r) return r;
    if (!sp->busy) return HVC_OK;
    DeviceGuard g(c->device);
    const hipError_t e = wait_event(sp->has_down ? sp->dn1 : sp->k1); // (polls and sleeps: the caller's other threads keep their CPUs)
    sp->busy = false;
    sp->timed = e == hipSuccess;
    if (e != hipSuccess) return fail_hip(c, e);
    return HVC_OK;
} HVC_ABI_CATCH

int hvc_slot_query(hvc_ctx *c, int slot, int *done) try {
    hvc_ctx::Slot *sp = nullptr;
    int r = slot_of(c, slot, &sp);
    if (r) return r;
    if (!done) return HVC_E_INVALID_ARG;
    *done = 1;
    if (!sp->busy) return HVC_OK;
    DeviceGuard g(c->device);
    const hipError_t e = hipEventQuery(sp->has_down ? sp->dn1 : sp->k1);
    if (e == hipErrorNotReady) {
        *done = 0;
        return HVC_OK;
    }
    if (e != hipSuccess) return fail_hip(c, e); // (hvc_wait will report it too, and free the slot)
    return HVC_OK;
} HVC_ABI_CATCH

int hvc_slot_last_stats(hvc_ctx *c, int slot, hvc_slot_stats *st) try {
    hvc_ctx::Slot *sp = nullptr;
    int r = slot_of(c, slot, &sp);
    if (r) return r;
    if (!st) return HVC_E_INVALID_ARG;
    std::memset(st, 0, sizeof *st);
    if (sp->busy || !sp->timed) return HVC_E_INVALID_ARG; // nothing completed in this slot yet (or it is still in flight)
    DeviceGuard g(c->device);
    float ms = 0;
    HIPCHK(c, hipEventElapsedTime(&ms, sp->up0, sp->up1));
    st->h2d_ms = ms;
    HIPCHK(c, hipEventElapsedTime(&ms, sp->k0, sp->k1));
    st->kernel_ms = ms;
    if (sp->has_down) {
        HIPCHK(c, hipEventElapsedTime(&ms, sp->dn0, sp->dn1));
        st->d2h_ms = ms;
    }
    st->h2d_bytes = sp->h2d_bytes;
    st->d2h_bytes = sp->d2h_bytes;
    return HVC_OK;
} HVC_ABI_CATCH
