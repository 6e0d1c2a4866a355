// mem_ubench.hip -- memory-pattern ceilings for the decode kernel's traffic shape on gfx950:
// 128 B read + 64 B written per lane ("block"), 2.4 GB per launch, almost no ALU work.
//   A  lane-strided 16 B loads (8 per lane, 128 B apart across lanes)  + 8 x 8 B row stores   (= K1 today)
//   B  wave-coalesced 16 B loads (lane i reads base + j*1024 + i*16)     + 8 x 8 B row stores
//   C  wave-coalesced loads + wave-coalesced 16 B stores (4 per lane)     (ideal 2:1 copy)
//   D  loads of A only (result folded into one dword per lane)            E  stores of A only
// Measurement tool only -- not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

constexpr int BW = 240, BH = 136;              // one 1080p luma plane worth of blocks per "plane"
constexpr size_t STRIDE = BW * 8;

template <int MODE>
__global__ __launch_bounds__(256) void k(const uint4 *__restrict__ in, unsigned char *__restrict__ out, unsigned *sink) {
    const size_t plane = blockIdx.y;
    const int b = blockIdx.x * 256 + threadIdx.x;       // block index in plane
    if (b >= BW * BH) return;
    const uint4 *src = in + plane * (size_t)(BW * BH) * 8;
    uint4 v[8];
    if (MODE == 0 || MODE == 3 || MODE == 6) {
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = src[(size_t)b * 8 + j];
    } else if (MODE == 5 || MODE == 7) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            typedef unsigned u4v __attribute__((ext_vector_type(4)));
            u4v t = __builtin_nontemporal_load(reinterpret_cast<const u4v *>(&src[(size_t)b * 8 + j]));
            v[j] = make_uint4(t.x, t.y, t.z, t.w);
        }
    } else if (MODE == 1 || MODE == 2) {
        const int wave_base = (b & ~63) * 8, l = b & 63;
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = src[(size_t)wave_base + j * 64 + l];
    } else {
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = make_uint4(b, j, plane, 7);
    }
    unsigned char *dst = out + plane * (STRIDE * BH * 8);
    if (MODE == 5 || MODE == 6) {
        const int by = b / BW, bx = b - by * BW;
        unsigned char *p = dst + (size_t)by * 8 * STRIDE + bx * 8;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            typedef unsigned u2v __attribute__((ext_vector_type(2)));
            u2v t = {v[j].x ^ v[j].z, v[j].y ^ v[j].w};
            __builtin_nontemporal_store(t, reinterpret_cast<u2v *>(p + j * STRIDE));
        }
    } else if (MODE == 0 || MODE == 1 || MODE == 4 || MODE == 7) {
        const int by = b / BW, bx = b - by * BW;
        unsigned char *p = dst + (size_t)by * 8 * STRIDE + bx * 8;
#pragma unroll
        for (int j = 0; j < 8; j++) *reinterpret_cast<uint2 *>(p + j * STRIDE) = make_uint2(v[j].x ^ v[j].z, v[j].y ^ v[j].w);
    } else if (MODE == 2) {
        uint4 *o = reinterpret_cast<uint4 *>(dst) + (size_t)(b & ~63) * 4 + (b & 63);
#pragma unroll
        for (int j = 0; j < 4; j++) o[j * 64] = make_uint4(v[2 * j].x ^ v[2 * j + 1].x, v[2 * j].y ^ v[2 * j + 1].y, v[2 * j].z ^ v[2 * j + 1].z, v[2 * j].w ^ v[2 * j + 1].w);
    } else {
        unsigned a = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) a ^= v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
        if (a == 0x12345678u) sink[0] = a;
    }
}

template <int MODE>
double run(const uint4 *in, unsigned char *out, unsigned *sink, int planes, int reps) {
    dim3 grid((BW * BH + 255) / 256, planes);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 5; i++) hipLaunchKernelGGL(k<MODE>, grid, dim3(256), 0, 0, in, out, sink);
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL(k<MODE>, grid, dim3(256), 0, 0, in, out, sink);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main() {
    const int planes = 384;                       // 384 x 32640 blocks x 192 B = 2.4 GB
    const size_t nblk = (size_t)planes * BW * BH;
    uint4 *in;
    unsigned char *out;
    unsigned *sink;
    CHECK(hipMalloc(&in, nblk * 128));
    CHECK(hipMalloc(&out, nblk * 64));
    CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(in, 1, nblk * 128));
    CHECK(hipMemset(out, 0, nblk * 64));
    const char *names[8] = {"A strided loads + row stores (K1 pattern)", "B coalesced loads + row stores",
                            "C coalesced loads + coalesced 16B stores", "D strided loads only", "E row stores only",
                            "F = A with nt loads + nt stores", "G = A with nt stores", "H = A with nt loads"};
    double bytes[8] = {192.0, 192.0, 192.0, 128.0, 64.0, 192.0, 192.0, 192.0};
    double ms[8];
    for (int rep = 0; rep < 2; rep++) {
        ms[0] = run<0>(in, out, sink, planes, 30);
        ms[1] = run<1>(in, out, sink, planes, 30);
        ms[2] = run<2>(in, out, sink, planes, 30);
        ms[3] = run<3>(in, out, sink, planes, 30);
        ms[4] = run<4>(in, out, sink, planes, 30);
        ms[5] = run<5>(in, out, sink, planes, 30);
        ms[6] = run<6>(in, out, sink, planes, 30);
        ms[7] = run<7>(in, out, sink, planes, 30);
    }
    for (int i = 0; i < 8; i++)
        printf("%-48s %8.4f ms  %8.1f GB/s\n", names[i], ms[i], nblk * bytes[i] / (ms[i] * 1e-3) / 1e9);
    return 0;
}
