// mem_ubench2.hip -- second round of memory-shape experiments for K1 (pure traffic, no arithmetic):
// which workgroup -> tile mapping / launch size gets closest to the chip's copy ceiling for
// "128 B read + 64 B written per block".  Sizes: 9.6 GB per launch like bench.py.
//   P  plain float4 copy, 1:1 (the guide's 6.29 TB/s reference), nt stores and plain
//   A  K1's shape: lane-strided 16 B loads + 8 x 8 B nt row stores, linear tile order (= shipped)
//   X  same, XCD-contiguous order: workgroup L serves tile (L % 8) * (T / 8) + L / 8
//   R  same, frame-major reversed nesting: grid.x = frame, grid.y = tile (neighbouring WGs touch
//      the same tile of different frames)
//   W  same, two tiles per workgroup (loop), half the workgroups
// Measurement tool only -- not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

constexpr int BW = 240, BH = 136, NBLK = BW * BH; // luma-plane sized "frames": 32640 blocks
constexpr int TILES = (NBLK + 255) / 256;          // 128 tiles (the last one partial: 32640 = 127.5 * 256)
constexpr size_t STRIDE = BW * 8;

typedef unsigned u2v __attribute__((ext_vector_type(2)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void do_tile(const uint4 *in, unsigned char *out, int frame, int tile, int lane) {
    const int b = tile * 256 + lane;
    if (b >= NBLK) return;
    const uint4 *src = in + ((size_t)frame * NBLK + b) * 8;
    uint4 v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = src[j];
    const int by = b / BW, bx = b - by * BW;
    unsigned char *p = out + (size_t)frame * (STRIDE * BH * 8) + (size_t)by * 8 * STRIDE + bx * 8;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        u2v t = {v[j].x ^ v[j].z, v[j].y ^ v[j].w};
        __builtin_nontemporal_store(t, reinterpret_cast<u2v *>(p + j * STRIDE));
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void k(const uint4 *__restrict__ in, unsigned char *__restrict__ out, int frames) {
    const int lane = threadIdx.x;
    if (MODE == 0) { // A: grid (TILES, frames)
        do_tile(in, out, blockIdx.y, blockIdx.x, lane);
    } else if (MODE == 1) { // X: 1-D grid, XCD-contiguous
        const unsigned total = (unsigned)TILES * frames, L = blockIdx.x;
        const unsigned per = total / 8; // total is a multiple of 8 here
        const unsigned t = (L % 8) * per + L / 8;
        do_tile(in, out, t / TILES, t % TILES, lane);
    } else if (MODE == 2) { // R: grid (frames, TILES)
        do_tile(in, out, blockIdx.x, blockIdx.y, lane);
    } else if (MODE == 3) { // W: two tiles per workgroup
        do_tile(in, out, blockIdx.y, 2 * blockIdx.x, lane);
        do_tile(in, out, blockIdx.y, 2 * blockIdx.x + 1, lane);
    }
}

// S: K1's loads, but the pixel rows leave as 16-byte nt stores, 1 KiB contiguous per wave instruction
// (what an LDS transpose across the workgroup's four waves would produce; data is arbitrary here so
// no LDS is needed).  Plane width 256 blocks so that a tile is exactly one block row; A256 = shape A
// on the same geometry.
template <int MODE>
__global__ __launch_bounds__(256) void ks(const uint4 *__restrict__ in, unsigned char *__restrict__ out) {
    const int lane = threadIdx.x, tile = blockIdx.x;
    const size_t frame = blockIdx.y;
    constexpr int TPF = 128; // tiles (block rows) per frame
    const uint4 *src = in + ((frame * TPF + tile) * 256 + lane) * 8;
    uint4 v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = src[j];
    unsigned char *base = out + (frame * TPF + tile) * (size_t)(2048 * 8);
    if (MODE == 0) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            u2v t = {v[j].x ^ v[j].z, v[j].y ^ v[j].w};
            __builtin_nontemporal_store(t, reinterpret_cast<u2v *>(base + j * 2048 + lane * 8));
        }
    } else {
        const int w = lane >> 6, l = lane & 63;
#pragma unroll
        for (int k = 0; k < 4; k++) { // rows 2w, 2w + 1; two 1 KiB halves each
            u4v t = {v[2 * k].x ^ v[2 * k + 1].x, v[2 * k].y ^ v[2 * k + 1].y, v[2 * k].z ^ v[2 * k + 1].z, v[2 * k].w ^ v[2 * k + 1].w};
            __builtin_nontemporal_store(t, reinterpret_cast<u4v *>(base + (2 * w + (k >> 1)) * 2048 + (k & 1) * 1024 + l * 16));
        }
    }
}

template <bool NT>
__global__ __launch_bounds__(256) void kcopy(const u4v *__restrict__ in, u4v *__restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 * 4 + threadIdx.x;
#pragma unroll
    for (int j = 0; j < 4; j++, i += 256)
        if (i < n) {
            u4v t = in[i];
            if (NT) __builtin_nontemporal_store(t, out + i); else out[i] = t;
        }
}

template <class F>
double timeit(F launch, int reps) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 8; i++) launch();
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; i++) launch();
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main(int argc, char **argv) {
    const int frames = argc > 1 ? atoi(argv[1]) : 1536; // 1536 x 32640 x 192 B = 9.6 GB
    const size_t nblk = (size_t)frames * NBLK;
    uint4 *in;
    unsigned char *out;
    CHECK(hipMalloc(&in, nblk * 128));
    CHECK(hipMalloc(&out, nblk * 128)); // also the destination of the 1:1 copy
    CHECK(hipMemset(in, 1, nblk * 128));
    CHECK(hipMemset(out, 0, nblk * 128));
    const int reps = 20;
    for (int rep = 0; rep < 2; rep++) {
        const size_t n16 = nblk * 8;
        double c0 = timeit([&] { hipLaunchKernelGGL(kcopy<false>, dim3((unsigned)((n16 + 1023) / 1024)), dim3(256), 0, 0, (const u4v *)in, (u4v *)out, n16); }, reps);
        double c1 = timeit([&] { hipLaunchKernelGGL(kcopy<true>, dim3((unsigned)((n16 + 1023) / 1024)), dim3(256), 0, 0, (const u4v *)in, (u4v *)out, n16); }, reps);
        double a = timeit([&] { hipLaunchKernelGGL(k<0>, dim3(TILES, frames), dim3(256), 0, 0, in, out, frames); }, reps);
        double x = timeit([&] { hipLaunchKernelGGL(k<1>, dim3((unsigned)TILES * frames), dim3(256), 0, 0, in, out, frames); }, reps);
        double r = timeit([&] { hipLaunchKernelGGL(k<2>, dim3(frames, TILES), dim3(256), 0, 0, in, out, frames); }, reps);
        double w = timeit([&] { hipLaunchKernelGGL(k<3>, dim3(TILES / 2, frames), dim3(256), 0, 0, in, out, frames); }, reps);
        const int f256 = (int)(nblk / (128 * 256));
        double a256 = timeit([&] { hipLaunchKernelGGL(ks<0>, dim3(128, f256), dim3(256), 0, 0, in, out); }, reps);
        double s256 = timeit([&] { hipLaunchKernelGGL(ks<1>, dim3(128, f256), dim3(256), 0, 0, in, out); }, reps);
        if (rep == 1) {
            printf("%-58s %8.4f ms %8.1f GB/s\n", "A256 K1 shape, 256-block-wide planes", a256, (double)f256 * 128 * 256 * 192.0 / (a256 * 1e-3) / 1e9);
            printf("%-58s %8.4f ms %8.1f GB/s\n", "S256 same loads, 16 B nt stores 1 KiB per wave instruction", s256, (double)f256 * 128 * 256 * 192.0 / (s256 * 1e-3) / 1e9);
        }
        if (rep == 1) {
            printf("frames %d (%.1f GB of 2:1 traffic)\n", frames, nblk * 192.0 / 1e9);
            printf("%-58s %8.4f ms %8.1f GB/s\n", "P  float4 copy 1:1, plain stores", c0, nblk * 256.0 / (c0 * 1e-3) / 1e9);
            printf("%-58s %8.4f ms %8.1f GB/s\n", "P' float4 copy 1:1, nt stores", c1, nblk * 256.0 / (c1 * 1e-3) / 1e9);
            printf("%-58s %8.4f ms %8.1f GB/s\n", "A  K1 shape, linear tile order (shipped)", a, nblk * 192.0 / (a * 1e-3) / 1e9);
            printf("%-58s %8.4f ms %8.1f GB/s\n", "X  K1 shape, XCD-contiguous tile order", x, nblk * 192.0 / (x * 1e-3) / 1e9);
            printf("%-58s %8.4f ms %8.1f GB/s\n", "R  K1 shape, frame index fastest", r, nblk * 192.0 / (r * 1e-3) / 1e9);
            printf("%-58s %8.4f ms %8.1f GB/s\n", "W  K1 shape, two tiles per workgroup", w, nblk * 192.0 / (w * 1e-3) / 1e9);
        }
    }
    return 0;
}
