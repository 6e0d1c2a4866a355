// store_ubench.hip -- write-pattern ceilings for the ENCODE kernel's output (128 B per lane):
//   S  lane-strided 16 B stores (8 per lane, 128 B apart across lanes)   (= k_encode before the LDS transpose)
//   T  wave-coalesced 16 B stores (lane i writes base + j*1024 + i*16)
//   U  like T, with the data moved through an XOR-swizzled LDS transpose first (what k_encode does)
// plus the 64 B/lane row loads, so the traffic is the encoder's 64 B in + 128 B out.  Tool only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
constexpr int BW = 480, BH = 270;
constexpr size_t STRIDE = BW * 8;

template <int MODE>
__global__ __launch_bounds__(256) void k(const unsigned char *__restrict__ in, uint4 *__restrict__ out) {
    __shared__ uint4 lds[4][512];
    const size_t plane = blockIdx.y;
    const int b = blockIdx.x * 256 + threadIdx.x;
    const bool active = b < BW * BH;
    const int bb = active ? b : BW * BH - 1;
    const int by = bb / BW, bx = bb - by * BW;
    const unsigned char *p = in + plane * (STRIDE * BH * 8) + (size_t)by * 8 * STRIDE + bx * 8;
    uint2 r[8];
#pragma unroll
    for (int j = 0; j < 8; j++) r[j] = *reinterpret_cast<const uint2 *>(p + j * STRIDE);
    uint4 v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = make_uint4(r[j].x, r[j].y, r[j].x ^ j, r[j].y + j);
    uint4 *dst = out + plane * (size_t)(BW * BH) * 8;
    if (MODE == 0) {
        if (active)
#pragma unroll
            for (int j = 0; j < 8; j++) dst[(size_t)bb * 8 + j] = v[j];
    } else if (MODE == 1) {
        const int wave_base = (bb & ~63) * 8, l = bb & 63;
        if (active)
#pragma unroll
            for (int j = 0; j < 8; j++) dst[(size_t)wave_base + j * 64 + l] = v[j];
    } else {
        const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
#pragma unroll
        for (int c = 0; c < 8; c++) lds[w][l * 8 + (c ^ (l & 7))] = v[c];
        const int wave_base = (bb & ~63) * 8;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int blk = 8 * j + (l >> 3), ch = l & 7;
            const uint4 t = lds[w][blk * 8 + (ch ^ (blk & 7))];
            if ((blockIdx.x * 256 + (threadIdx.x & ~63) + blk) < BW * BH) dst[(size_t)wave_base + j * 64 + l] = t;
        }
    }
}

template <int MODE>
double run(const unsigned char *in, uint4 *out, int planes, int reps) {
    dim3 grid((BW * BH + 255) / 256, planes);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 5; i++) hipLaunchKernelGGL(k<MODE>, grid, dim3(256), 0, 0, in, out);
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL(k<MODE>, grid, dim3(256), 0, 0, in, out);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main() {
    const int planes = 96; // 96 x 129600 blocks x 192 B = 2.39 GB
    const size_t nblk = (size_t)planes * BW * BH;
    unsigned char *in;
    uint4 *out;
    CHECK(hipMalloc(&in, nblk * 64));
    CHECK(hipMalloc(&out, nblk * 128));
    CHECK(hipMemset(in, 3, nblk * 64));
    const char *names[3] = {"S lane-strided 16B stores", "T wave-coalesced 16B stores", "U LDS transpose + coalesced 16B stores"};
    double ms[3];
    for (int rep = 0; rep < 2; rep++) {
        ms[0] = run<0>(in, out, planes, 30);
        ms[1] = run<1>(in, out, planes, 30);
        ms[2] = run<2>(in, out, planes, 30);
    }
    for (int i = 0; i < 3; i++) printf("%-44s %8.4f ms  %8.1f GB/s\n", names[i], ms[i], nblk * 192.0 / (ms[i] * 1e-3) / 1e9);
    return 0;
}
