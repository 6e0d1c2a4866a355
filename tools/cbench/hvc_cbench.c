/* hvc_cbench.c -- the block-transform path driven from plain C through include/hvc_jpeg.h only
 * (no Python, no PyTorch, no HIP headers): what a cgo / OCaml-ctypes / JNI caller does.
 *
 *   gcc -std=c99 -O2 -I include tools/cbench/hvc_cbench.c -o /tmp/hvc_cbench \
 *       video-coding_amd/libhvc_jpeg.so -Wl,-rpath,$PWD/video-coding_amd
 *   /tmp/hvc_cbench [frames=64] [steps=20] [width=1920] [height=1080]
 *
 * Frames of LCG pixels -> hvc_encode_frames (host buffers) -> coefficient records uploaded once with
 * hvc_device_alloc / hvc_memcpy_h2d -> `steps` x hvc_decode_frames on device memory, timed with
 * hvc_timer_* -> pixels downloaded; prints Mpixel/s, algorithmic GB/s and the CRC-32 of frame 0's
 * pixel record (tests/test_gpu_cbench.py recomputes it through the Python harness and the oracle).
 * Then the same frames through the ASYNCHRONOUS seam the way a single-threaded caller with its own entropy reader uses it
 * (the OCaml patch's Decoder.decode_frames_gpu): pinned slot buffers from hvc_host_alloc; for every batch, "read" it into the
 * next slot's pinned record (here: a memcpy) while the earlier submissions are in flight, hvc_decode_frames_submit, and
 * hvc_wait on the slot only when its turn comes round again.  Prints the rate over the whole loop and the CRC-32 of frame 0 of
 * the last batch's downloaded pixel record, which must equal the first one. */
#define _POSIX_C_SOURCE 199309L
#include <time.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "hvc_jpeg.h"

#define CHECK(call)                                                                   \
    do {                                                                              \
        int rc_ = (call);                                                             \
        if (rc_ != HVC_OK) {                                                          \
            fprintf(stderr, "%s -> %d (%s)\n", #call, rc_, hvc_strerror(rc_));        \
            return 1;                                                                 \
        }                                                                             \
    } while (0)

static uint32_t crc32_bytes(const uint8_t *p, size_t n) {
    uint32_t c = 0xffffffffu;
    size_t i;
    int k;
    for (i = 0; i < n; i++) {
        c ^= p[i];
        for (k = 0; k < 8; k++) c = (c >> 1) ^ (0xedb88320u & (0u - (c & 1u)));
    }
    return ~c;
}

int main(int argc, char **argv) {
    const int frames = argc > 1 ? atoi(argv[1]) : 64, steps = argc > 2 ? atoi(argv[2]) : 20;
    const int width = argc > 3 ? atoi(argv[3]) : 1920, height = argc > 4 ? atoi(argv[4]) : 1080;
    hvc_jpeg_info info;
    hvc_ctx *ctx = NULL;
    uint8_t *pix, *d_pix = NULL;
    int16_t *coefs, *d_coefs = NULL;
    size_t i, blocks = 0;
    uint32_t lcg = 12345u;
    float ms = 0.0f;
    int f, s;

    if (frames < 1 || steps < 1) return 2;
    CHECK(hvc_jpeg_encoder_layout(width, height, 420, 75, &info)); /* geometry + Quant_tables.scale 75 */
    for (f = 0; f < info.n_comp; f++) blocks += (size_t)info.layout[f].blocks_w * info.layout[f].blocks_h;
    CHECK(hvc_create(&ctx, 0));
    pix = (uint8_t *)malloc(info.pixel_bytes * (size_t)frames);
    coefs = (int16_t *)malloc(info.coef_count * sizeof(int16_t) * (size_t)frames);
    if (!pix || !coefs) return 3;
    for (i = 0; i < info.pixel_bytes * (size_t)frames; i++) { /* smooth-ish content: slow ramp + LCG noise */
        lcg = lcg * 1664525u + 1013904223u;
        pix[i] = (uint8_t)(((i >> 3) & 0x7f) + ((lcg >> 24) & 0x3f));
    }
    CHECK(hvc_encode_frames(ctx, pix, info.pixel_bytes, &info.qtabs[0][0], info.n_qtabs, info.layout, info.n_comp,
                            frames, coefs, info.coef_count, HVC_MEM_HOST));
    CHECK(hvc_device_alloc(ctx, info.coef_count * sizeof(int16_t) * (size_t)frames, (void **)&d_coefs));
    CHECK(hvc_device_alloc(ctx, info.pixel_bytes * (size_t)frames, (void **)&d_pix));
    CHECK(hvc_memcpy_h2d(ctx, d_coefs, coefs, info.coef_count * sizeof(int16_t) * (size_t)frames));
    for (s = 0; s < 5; s++) /* warm-up */
        CHECK(hvc_decode_frames(ctx, d_coefs, info.coef_count, &info.qtabs[0][0], info.n_qtabs, info.layout,
                                info.n_comp, frames, d_pix, info.pixel_bytes, HVC_MEM_DEVICE));
    CHECK(hvc_synchronize(ctx));
    CHECK(hvc_timer_begin(ctx));
    for (s = 0; s < steps; s++)
        CHECK(hvc_decode_frames(ctx, d_coefs, info.coef_count, &info.qtabs[0][0], info.n_qtabs, info.layout,
                                info.n_comp, frames, d_pix, info.pixel_bytes, HVC_MEM_DEVICE));
    CHECK(hvc_timer_end(ctx, &ms));
    memset(pix, 0, info.pixel_bytes);
    CHECK(hvc_memcpy_d2h(ctx, pix, d_pix, info.pixel_bytes));
    printf("{\"frames\": %d, \"steps\": %d, \"width\": %d, \"height\": %d, \"ms_per_step\": %.4f, "
           "\"Mpixel_s\": %.1f, \"algorithmic_GBps\": %.1f, \"crc32_frame0\": %u}\n",
           frames, steps, width, height, ms / steps, (double)frames * width * height / (ms / steps * 1e-3) / 1e6,
           (double)frames * (double)blocks * 192.0 / (ms / steps * 1e-3) / 1e9, crc32_bytes(pix, info.pixel_bytes));
    CHECK(hvc_device_free(ctx, d_coefs));
    CHECK(hvc_device_free(ctx, d_pix));
    {
        const size_t cbytes = info.coef_count * sizeof(int16_t) * (size_t)frames, pbytes = info.pixel_bytes * (size_t)frames;
        int16_t *pin_c[HVC_SLOTS];
        uint8_t *pin_p[HVC_SLOTS];
        const int batches = steps < HVC_SLOTS + 1 ? HVC_SLOTS + 1 : steps;
        struct timespec t0, t1;
        double wall_ms;
        hvc_slot_stats st;
        int k;
        for (k = 0; k < HVC_SLOTS; k++) {
            CHECK(hvc_host_alloc(ctx, cbytes, (void **)&pin_c[k]));
            CHECK(hvc_host_alloc(ctx, pbytes, (void **)&pin_p[k]));
            memset(pin_p[k], 0, pbytes);
        }
        clock_gettime(CLOCK_MONOTONIC, &t0);
        for (k = 0; k < batches; k++) {
            const int slot = k % HVC_SLOTS;
            CHECK(hvc_wait(ctx, slot));            /* batch k - HVC_SLOTS: its pixels are in pin_p[slot] now (idle slot: returns at once) */
            memcpy(pin_c[slot], coefs, cbytes);    /* the caller's reader fills the pinned record -- batches k - 1, k - 2, ... are in flight */
            CHECK(hvc_decode_frames_submit(ctx, slot, pin_c[slot], info.coef_count, &info.qtabs[0][0], info.n_qtabs, info.layout,
                                           info.n_comp, frames, pin_p[slot], info.pixel_bytes, HVC_MEM_HOST));
        }
        for (k = 0; k < HVC_SLOTS; k++) CHECK(hvc_wait(ctx, k));
        clock_gettime(CLOCK_MONOTONIC, &t1);
        wall_ms = (double)(t1.tv_sec - t0.tv_sec) * 1e3 + (double)(t1.tv_nsec - t0.tv_nsec) * 1e-6;
        CHECK(hvc_slot_last_stats(ctx, (batches - 1) % HVC_SLOTS, &st));
        printf("{\"async_batches\": %d, \"slots\": %d, \"async_Mpixel_s\": %.1f, \"last_h2d_GBps\": %.1f, \"last_d2h_GBps\": %.1f, "
               "\"async_crc32_frame0\": %u}\n", batches, (int)HVC_SLOTS, (double)batches * frames * width * height / (wall_ms * 1e-3) / 1e6,
               (double)st.h2d_bytes / (st.h2d_ms * 1e-3) / 1e9, (double)st.d2h_bytes / (st.d2h_ms * 1e-3) / 1e9,
               crc32_bytes(pin_p[(batches - 1) % HVC_SLOTS], info.pixel_bytes));
        for (k = 0; k < HVC_SLOTS; k++) {
            CHECK(hvc_host_free(ctx, pin_c[k]));
            CHECK(hvc_host_free(ctx, pin_p[k]));
        }
    }
    hvc_destroy(ctx);
    free(pix);
    free(coefs);
    return 0;
}
