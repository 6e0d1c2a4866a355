// wgsize_ubench.hip -- is the granule effect of inflight_ubench (the ceiling of a read:write mix falls as a lane moves more
// bytes) a matter of the bytes per LANE or of the bytes one WORKGROUP's region spans?  The same ideal shape (every wave
// instruction 1 KiB contiguous, a wave's pieces contiguous, a workgroup's waves side by side) at K1's, K3's and the fused
// kernel's bytes per lane, in workgroups of 64 / 128 / 256 / 512 / 1024 lanes; and the small granule (2 + 1 pieces) in
// the same workgroup sizes.  Measurement tool only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef unsigned u4v __attribute__((ext_vector_type(4)));

template <int R, int W, int WGS>
__global__ __launch_bounds__(WGS) void k(const u4v *__restrict__ in, u4v *__restrict__ out, size_t waves) {
    const size_t wave = (size_t)blockIdx.x * (WGS / 64) + (threadIdx.x >> 6);
    if (wave >= waves) return;
    const int l = threadIdx.x & 63;
    const u4v *src = in + wave * (size_t)(R * 64) + l;
    u4v r[R];
#pragma unroll
    for (int j = 0; j < R; j++) r[j] = src[j * 64];
    u4v acc = r[0];
#pragma unroll
    for (int j = 1; j < R; j++) acc ^= r[j];
    u4v *dst = out + wave * (size_t)(W * 64) + l;
#pragma unroll
    for (int j = 0; j < W; j++) {
        u4v t = acc;
        t.x += (unsigned)j;
        __builtin_nontemporal_store(t, dst + j * 64);
    }
}

template <class F>
double timeit(F launch, int reps) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; i++) launch();
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; i++) launch();
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

static const size_t TOTAL = 9600ull * 1000000ull;
static u4v *A, *B;

template <int R, int W, int WGS>
double one() {
    const size_t waves = TOTAL / ((size_t)(R + W) * 1024);
    const unsigned grid = (unsigned)((waves + WGS / 64 - 1) / (WGS / 64));
    const double ms = timeit([&] { hipLaunchKernelGGL((k<R, W, WGS>), dim3(grid), dim3(WGS), 0, 0, A, B, waves); }, 15);
    return (double)waves * (R + W) * 1024 / (ms * 1e-3) / 8e12 * 100;
}
template <int R, int W>
void row(bool pr, const char *name) {
    const double a = one<R, W, 64>(), b = one<R, W, 128>(), c = one<R, W, 256>(), d = one<R, W, 512>(), e = one<R, W, 1024>();
    if (pr) printf("%-36s workgroup of 64: %5.1f   128: %5.1f   256: %5.1f   512: %5.1f   1024: %5.1f   (%% of 8 TB/s)\n", name, a, b, c, d, e);
}

int main() {
    CHECK(hipMalloc(&A, TOTAL));
    CHECK(hipMalloc(&B, TOTAL));
    CHECK(hipMemset(A, 1, TOTAL));
    CHECK(hipMemset(B, 0, TOTAL));
    for (int pass = 0; pass < 3; pass++) {
        const bool pr = pass > 0;
        if (pr) printf("-- pass %d\n", pass);
        row<2, 1>(pr, "2:1, 32 B in + 16 B out per lane");
        row<8, 4>(pr, "2:1, 128 B in + 64 B out (K1)");
        row<1, 2>(pr, "1:2, 16 B in + 32 B out per lane");
        row<4, 8>(pr, "1:2, 64 B in + 128 B out (K3)");
        row<8, 8>(pr, "1:1, 128 B in + 128 B out (fused)");
    }
    return 0;
}
