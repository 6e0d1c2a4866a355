// shape_ubench.hip -- which half of a kernel's access shape costs it the distance to the mix ceiling?
// Pure traffic, no arithmetic; hybrids of the REAL load / store address patterns of K3 (k_encode) and of the fused
// 4:4:4 kernel's two halves with the IDEAL pattern (every wave instruction 1 KiB contiguous, 16 B per lane):
//
//   K3 (1:2, 4K 4:2:0 luma-sized planes 480 x 270 blocks, 256-block linear tiles, 64 B in + 128 B out per block)
//     e_real   real loads (8 x 8 B per lane, a wave instruction = 512 B of a pixel row) + real stores (8 x 16 B, 1 KiB runs)
//     e_ideal  ideal loads (4 x 16 B per lane, the wave's 4 KiB contiguous) + the same stores
//     e_pair   two horizontally adjacent blocks per lane: 8 x 16 B loads, a wave instruction = 1 KiB of a pixel row
//     e_lds    e_real with the stores' LDS round trip (what the shipped kernel does)
//   fused chroma half (1:2, 1080p chroma planes 120 x 68 blocks -> 1920-wide 4:4:4 rows; 128 B in + 256 B out per block)
//     c_real   K1-shape loads (8 x 16 B per lane at 128 B stride) + 16 x 16 B nt stores per lane, rows 1920 B apart
//     c_lideal ideal loads + the real stores
//     c_sideal real loads + ideal stores (16 x 16 B per lane, wave-contiguous)
//   fused luma half / K1 (2:1, 240 x 135 blocks, rows 1920 B apart)
//     y_real   K1-shape loads + 8 x 8 B nt row stores
//     y_lideal ideal loads + real stores
//     y_sideal real loads + ideal stores
//   mix<R,W> of mem_ubench3 at 1:2 and 2:1 for the ceiling on the same box.
// Measurement tool only -- not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

typedef unsigned u2v __attribute__((ext_vector_type(2)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------- K3
constexpr int EBW = 480, EBH = 270, ENB = EBW * EBH;   // blocks per plane
constexpr int ETILES = (ENB + 255) / 256;
constexpr size_t ESTRIDE = (size_t)EBW * 8;            // bytes per pixel row

template <int MODE> // 0 real, 1 ideal loads, 2 pair loads, 3 real + LDS round trip
__global__ __launch_bounds__(256) void k_enc(const unsigned char *__restrict__ pix, u4v *__restrict__ coefs) {
    __shared__ u4v lds[MODE == 3 ? 4 : 1][MODE == 3 ? 512 : 1];
    const int lane = threadIdx.x, wv = lane >> 6, l = lane & 63;
    const size_t plane = blockIdx.y;
    const unsigned char *pp = pix + plane * (ESTRIDE * EBH * 8);
    u4v *cp = coefs + plane * ((size_t)ENB * 8);
    u4v acc[8];
    if (MODE == 2) {
        // a tile = 512 blocks; lane l of the workgroup owns blocks 2l, 2l + 1 (bw even: the same block row)
        const int b = blockIdx.x * 512 + 2 * lane;
        if (b >= ENB) return;
        const int by = b / EBW, bx = b - by * EBW;
        const unsigned char *p = pp + (size_t)by * 8 * ESTRIDE + (size_t)bx * 8;
        u4v r[8];
#pragma unroll
        for (int j = 0; j < 8; j++) r[j] = *reinterpret_cast<const u4v *>(p + j * ESTRIDE);
        // 2 x 128 B out per lane = 256 B contiguous; wave-contiguous 1 KiB runs (data arbitrary, so no transpose needed here)
        const int wave_b0 = blockIdx.x * 512 + (lane & ~63) * 2;
        u4v *dst = cp + (size_t)wave_b0 * 8;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            u4v t = r[j & 7];
            t.x += j;
            if (wave_b0 + (j * 64 + l) / 8 < ENB) __builtin_nontemporal_store(t, dst + j * 64 + l);
        }
        return;
    }
    const int b = blockIdx.x * 256 + lane;
    const bool active = b < ENB;
    const int bc = active ? b : ENB - 1;
    if (MODE == 1) {
        // ideal: the wave's 64 blocks x 64 B = 4 KiB as four wave-contiguous 1 KiB pieces (of the plane seen as a linear array)
        const u4v *src = reinterpret_cast<const u4v *>(pp) + ((size_t)blockIdx.x * 256 + (lane & ~63)) * 4;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const size_t at = (size_t)j * 64 + l;
            acc[j] = ((size_t)blockIdx.x * 256 + (lane & ~63)) * 4 + at < (size_t)ENB * 4 ? src[at] : u4v{0, 0, 0, 0};
            acc[j + 4] = acc[j];
        }
    } else {
        const int by = bc / EBW, bx = bc - by * EBW;
        const unsigned char *p = pp + (size_t)by * 8 * ESTRIDE + (size_t)bx * 8;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const u2v w = *reinterpret_cast<const u2v *>(p + j * ESTRIDE);
            acc[j] = u4v{w.x, w.y, w.x ^ 1u, w.y ^ 2u};
        }
    }
    if (MODE == 3) { // the shipped kernel's transpose: lane-major in, piece-major out
#pragma unroll
        for (int j = 0; j < 8; j++) lds[wv][l * 8 + (j ^ (l & 7))] = acc[j];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int blk = 8 * j + (l >> 3), ch = l & 7;
            acc[j] = lds[wv][blk * 8 + (ch ^ (blk & 7))];
        }
    }
    const int wave_b0 = blockIdx.x * 256 + (lane & ~63);
    u4v *dst = cp + (size_t)wave_b0 * 8;
#pragma unroll
    for (int j = 0; j < 8; j++)
        if (wave_b0 + 8 * j + (l >> 3) < ENB) __builtin_nontemporal_store(acc[j], dst + j * 64 + l);
}

// ------------------------------------------------------------------------------- fused 4:4:4, chroma half
constexpr int CBW = 120, CBH = 68, CW = 1920;          // source blocks; output rows of 1920 bytes, 16 rows per block row
constexpr int CTILE_W = 128, CTILE_H = 4;              // one workgroup: 128 x 4 blocks (512 lanes), 120 of 128 busy
template <int MODE> // 0 real, 1 ideal loads, 2 ideal stores
__global__ __launch_bounds__(512, 2) void k_chroma(const u4v *__restrict__ coefs, unsigned char *__restrict__ out) {
    const int lane = threadIdx.x, lx = lane & 127, ly = lane >> 7, l = lane & 63;
    const size_t plane = blockIdx.y;
    const int bx = lx, by = blockIdx.x * CTILE_H + ly;
    const bool active = bx < CBW && by < CBH;
    const u4v *cp = coefs + plane * ((size_t)CBW * CBH * 8);
    unsigned char *op = out + plane * ((size_t)CW * CBH * 16);
    u4v r[8];
    if (MODE == 1) { // ideal loads: the workgroup's 512 x 128 B as wave-contiguous 1 KiB pieces
        const u4v *src = cp + ((size_t)blockIdx.x * 512 + (lane & ~63)) * 8;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const size_t at = ((size_t)blockIdx.x * 512 + (lane & ~63)) * 8 + (size_t)j * 64 + l;
            r[j] = at < (size_t)CBW * CBH * 8 ? src[(size_t)j * 64 + l] : u4v{0, 0, 0, 0};
        }
    } else {
        const int bxc = bx < CBW ? bx : CBW - 1, byc = by < CBH ? by : CBH - 1;
        const u4v *src = cp + ((size_t)byc * CBW + bxc) * 8;
#pragma unroll
        for (int j = 0; j < 8; j++) r[j] = src[j];
    }
    if (MODE == 2) { // ideal stores: 16 x 16 B per lane, wave-contiguous
        u4v *dst = reinterpret_cast<u4v *>(op) + ((size_t)blockIdx.x * 512 + (lane & ~63)) * 16;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            u4v t = r[j & 7];
            t.x += j;
            if (((size_t)blockIdx.x * 512 + (lane & ~63)) * 16 + (size_t)j * 64 + l < (size_t)CW * CBH) // (16 B units: CW * CBH * 16 / 16)
                __builtin_nontemporal_store(t, dst + (size_t)j * 64 + l);
        }
        return;
    }
    if (!active) return;
    unsigned char *p = op + (size_t)by * 16 * CW + (size_t)bx * 16;
#pragma unroll
    for (int j = 0; j < 16; j++) {
        u4v t = r[j & 7];
        t.x += j;
        __builtin_nontemporal_store(t, reinterpret_cast<u4v *>(p + (size_t)j * CW));
    }
}

// Two lanes per chroma block ("each lane stores half the bytes"): the per-lane granule halves to 64 B in +
// 128 B out.  HALVES = 0: the pair sits side by side in the wave (lane 2k / 2k + 1 = block k's first / second 64 B of
// coefficients and the left / right 8 output columns: 16 x 8 B row stores, a wave instruction = 512 B of an output row) -- what a
// kernel that splits the passes between the two lanes and exchanges by DPP would do.  HALVES = 1: the two halves in different
// waves (each lane 4 x 16 B of its block and the upper or lower 8 output rows as 16 B stores, a wave instruction = 1 KiB of a
// row) -- the instruction shapes of the shipped kernel at half the granule.
template <int HALVES>
__global__ __launch_bounds__(512, 2) void k_chroma_pair(const u4v *__restrict__ coefs, unsigned char *__restrict__ out) {
    const int lane = threadIdx.x;
    const size_t plane = blockIdx.y;
    const u4v *cp = coefs + plane * ((size_t)CBW * CBH * 8);
    unsigned char *op = out + plane * ((size_t)CW * CBH * 16);
    int bx, by, half;
    if (HALVES == 0) {
        bx = (lane & 255) >> 1;
        half = lane & 1;
        by = blockIdx.x * 2 + (lane >> 8);
    } else {
        bx = lane & 127;
        by = blockIdx.x * 2 + ((lane >> 7) & 1);
        half = lane >> 8;
    }
    const bool active = bx < CBW && by < CBH;
    const int bxc = bx < CBW ? bx : CBW - 1, byc = by < CBH ? by : CBH - 1;
    const u4v *src = cp + ((size_t)byc * CBW + bxc) * 8 + half * 4;
    u4v r[4];
#pragma unroll
    for (int j = 0; j < 4; j++) r[j] = src[j];
    if (!active) return;
    if (HALVES == 0) {
        unsigned char *p = op + (size_t)by * 16 * CW + (size_t)bx * 16 + half * 8;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            u2v t = {r[j & 3].x + (unsigned)j, r[j & 3].y ^ r[j & 3].w};
            __builtin_nontemporal_store(t, reinterpret_cast<u2v *>(p + (size_t)j * CW));
        }
    } else {
        unsigned char *p = op + ((size_t)by * 16 + half * 8) * CW + (size_t)bx * 16;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            u4v t = r[j & 3];
            t.x += j;
            __builtin_nontemporal_store(t, reinterpret_cast<u4v *>(p + (size_t)j * CW));
        }
    }
}

// --------------------------------------------------------------------------------- fused luma half / K1
constexpr int YBW = 240, YBH = 135, YNB = YBW * YBH;
template <int MODE, int WGS> // 0 real, 1 ideal loads, 2 ideal stores
__global__ __launch_bounds__(WGS) void k_luma(const u4v *__restrict__ coefs, unsigned char *__restrict__ out) {
    const int lane = threadIdx.x, l = lane & 63;
    const size_t plane = blockIdx.y;
    const int b = blockIdx.x * WGS + lane;
    const bool active = b < YNB;
    const int bc = active ? b : YNB - 1;
    const u4v *cp = coefs + plane * ((size_t)YNB * 8);
    unsigned char *op = out + plane * ((size_t)CW * YBH * 8);
    u4v r[8];
    const size_t wave0 = (size_t)blockIdx.x * WGS + (lane & ~63);
    if (MODE == 1) {
#pragma unroll
        for (int j = 0; j < 8; j++) r[j] = wave0 * 8 + (size_t)j * 64 + l < (size_t)YNB * 8 ? cp[wave0 * 8 + (size_t)j * 64 + l] : u4v{0, 0, 0, 0};
    } else {
#pragma unroll
        for (int j = 0; j < 8; j++) r[j] = cp[(size_t)bc * 8 + j];
    }
    if (MODE == 2) { // 64 B per lane as 4 x 16 B wave-contiguous
        u4v *dst = reinterpret_cast<u4v *>(op) + wave0 * 4;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const u4v t = r[j] ^ r[j + 4];
            if (wave0 * 4 + (size_t)j * 64 + l < (size_t)YNB * 4) __builtin_nontemporal_store(t, dst + (size_t)j * 64 + l);
        }
        return;
    }
    if (!active) return;
    const int by = bc / YBW, bx = bc - by * YBW;
    unsigned char *p = op + (size_t)by * 8 * CW + (size_t)bx * 8;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const u2v t = {r[j].x ^ r[j].z, r[j].y ^ r[j].w};
        __builtin_nontemporal_store(t, reinterpret_cast<u2v *>(p + (size_t)j * CW));
    }
}

// ------------------------------------------------------------------------------------------ ceilings
template <int R, int W>
__global__ __launch_bounds__(256) void mix(const u4v *__restrict__ in, u4v *__restrict__ out, size_t n) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    u4v acc = {1u, 2u, 3u, 4u};
#pragma unroll
    for (int j = 0; j < R; j++) acc ^= in[(size_t)j * n + t];
#pragma unroll
    for (int j = 0; j < W; j++) {
        u4v o = acc;
        o.x += (unsigned)j;
        __builtin_nontemporal_store(o, out + (size_t)j * n + t);
    }
}

template <class F>
double timeit(F launch, int reps) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 5; i++) launch();
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; i++) launch();
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

static void line(const char *name, double ms, double bytes) {
    printf("%-58s %8.4f ms %8.1f GB/s  (%.1f %% of 8 TB/s)\n", name, ms, bytes / (ms * 1e-3) / 1e9, bytes / (ms * 1e-3) / 8e12 * 100);
}

int main() {
    const size_t total = 9600ull * 1000000ull; // ~bytes moved per launch
    u4v *a;
    unsigned char *b;
    CHECK(hipMalloc(&a, total));
    CHECK(hipMalloc(&b, total));
    CHECK(hipMemset(a, 1, total));
    CHECK(hipMemset(b, 0, total));
    const int reps = 20;
    for (int pass = 0; pass < 3; pass++) { // pass 0 warms up; passes 1 and 2 print (alternating order = same box, minutes apart)
        const bool pr = pass > 0;
        if (pr) printf("-- pass %d\n", pass);
        { // K3: planes of ENB blocks, 192 B each
            const int planes = (int)(total / ((size_t)ENB * 192));
            const double bytes = (double)planes * ENB * 192;
            double t0 = timeit([&] { hipLaunchKernelGGL(k_enc<0>, dim3(ETILES, planes), dim3(256), 0, 0, (const unsigned char *)a, (u4v *)b); }, reps);
            double t1 = timeit([&] { hipLaunchKernelGGL(k_enc<1>, dim3(ETILES, planes), dim3(256), 0, 0, (const unsigned char *)a, (u4v *)b); }, reps);
            double t2 = timeit([&] { hipLaunchKernelGGL(k_enc<2>, dim3((ENB + 511) / 512, planes), dim3(256), 0, 0, (const unsigned char *)a, (u4v *)b); }, reps);
            double t3 = timeit([&] { hipLaunchKernelGGL(k_enc<3>, dim3(ETILES, planes), dim3(256), 0, 0, (const unsigned char *)a, (u4v *)b); }, reps);
            const size_t n = total / 16 / 3;
            double tc = timeit([&] { hipLaunchKernelGGL((mix<1, 2>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, (const u4v *)a, (u4v *)b, n); }, reps);
            if (pr) {
                line("K3 e_real   8 x 8 B row loads + 1 KiB-run stores", t0, bytes);
                line("K3 e_lds    ... with the LDS round trip (shipped shape)", t3, bytes);
                line("K3 e_ideal  wave-contiguous loads + the same stores", t1, bytes);
                line("K3 e_pair   two blocks per lane: 8 x 16 B row loads", t2, bytes);
                line("mix 1:2 (ideal shape, nt stores)", tc, (double)n * 48);
            }
        }
        { // chroma half: planes of CBW x CBH blocks, 128 B in + 256 B out
            const int planes = (int)(total / ((size_t)CBW * CBH * 384));
            const double bytes = (double)planes * CBW * CBH * 384;
            const dim3 g((CBH + CTILE_H - 1) / CTILE_H, planes);
            double t0 = timeit([&] { hipLaunchKernelGGL(k_chroma<0>, g, dim3(512), 0, 0, (const u4v *)a, b); }, reps);
            double t1 = timeit([&] { hipLaunchKernelGGL(k_chroma<1>, g, dim3(512), 0, 0, (const u4v *)a, b); }, reps);
            double t2 = timeit([&] { hipLaunchKernelGGL(k_chroma<2>, g, dim3(512), 0, 0, (const u4v *)a, b); }, reps);
            const dim3 g2((CBH + 1) / 2, planes);
            double t3 = timeit([&] { hipLaunchKernelGGL(k_chroma_pair<0>, g2, dim3(512), 0, 0, (const u4v *)a, b); }, reps);
            double t4 = timeit([&] { hipLaunchKernelGGL(k_chroma_pair<1>, g2, dim3(512), 0, 0, (const u4v *)a, b); }, reps);
            if (pr) {
                line("444 chroma c_pair   two lanes per block side by side: 4 x 16 B loads + 16 x 8 B stores", t3, bytes);
                line("444 chroma c_halves two lanes per block in two waves: 4 x 16 B loads + 8 x 16 B stores", t4, bytes);
                line("444 chroma c_real   K1-shape loads + 16 x 16 B row stores", t0, bytes);
                line("444 chroma c_lideal ideal loads + the real stores", t1, bytes);
                line("444 chroma c_sideal real loads + ideal stores", t2, bytes);
            }
        }
        { // luma half
            const int planes = (int)(total / ((size_t)YNB * 192));
            const double bytes = (double)planes * YNB * 192;
            double t0 = timeit([&] { hipLaunchKernelGGL((k_luma<0, 256>), dim3((YNB + 255) / 256, planes), dim3(256), 0, 0, (const u4v *)a, b); }, reps);
            double t5 = timeit([&] { hipLaunchKernelGGL((k_luma<0, 512>), dim3((YNB + 511) / 512, planes), dim3(512), 0, 0, (const u4v *)a, b); }, reps);
            double t1 = timeit([&] { hipLaunchKernelGGL((k_luma<1, 256>), dim3((YNB + 255) / 256, planes), dim3(256), 0, 0, (const u4v *)a, b); }, reps);
            double t2 = timeit([&] { hipLaunchKernelGGL((k_luma<2, 256>), dim3((YNB + 255) / 256, planes), dim3(256), 0, 0, (const u4v *)a, b); }, reps);
            const size_t n = total / 16 / 3;
            double tc = timeit([&] { hipLaunchKernelGGL((mix<2, 1>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, (const u4v *)a, (u4v *)b, n); }, reps);
            if (pr) {
                line("444 luma y_real   K1-shape loads + 8 x 8 B row stores (256 lanes)", t0, bytes);
                line("444 luma y_real   ... in workgroups of 512 lanes", t5, bytes);
                line("444 luma y_lideal ideal loads + the real stores", t1, bytes);
                line("444 luma y_sideal real loads + ideal stores", t2, bytes);
                line("mix 2:1 (ideal shape, nt stores)", tc, (double)n * 48);
            }
        }
    }
    return 0;
}
