// inflight_ubench.hip -- how does the streaming ceiling of a read:write mix move with the bytes a wave has in flight?
// The ideal-shape mix of mem_ubench3 moves 16 B in + 32 B out per lane (1:2) and reaches 77 % of 8 TB/s; K3's own
// shape as pure traffic (64 B in + 128 B out per lane, every instruction still 1 KiB contiguous per wave) reaches 70 %
// on the same box (profiles/r03b_ab.txt).  Here: the same ideal shape with R x 16 B loaded up front and W x 16 B
// stored per lane, R:W = 2:1 and 1:2, at 8 ... 2 workgroups per CU (dynamic LDS) -- bytes in flight per CU =
// workgroups x 4 waves x R KiB.  Measurement tool only -- not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef unsigned u4v __attribute__((ext_vector_type(4)));

// workgroup g reads the R * 4 KiB at in + g * R * 256 pieces and writes the W * 4 KiB at out + g * W * 256 pieces;
// instruction j of a wave covers one contiguous KiB
// sh >= 0 (round 4): the workgroups do not take their regions in dispatch order (consecutive regions behind eight different
// XCDs' L2s) but every XCD takes runs of 2^sh consecutive regions (csrc/hvc_kernels.h xcd_work)
template <int R, int W>
__global__ __launch_bounds__(256) void k(const u4v *__restrict__ in, u4v *__restrict__ out, size_t groups, int sh) {
    extern __shared__ unsigned char dyn_lds[];
    unsigned id = blockIdx.x;
    if (sh >= 0) {
        const unsigned group = 8u << sh, total = gridDim.x;
        if (id < total - total % group) {
            const unsigned kk = id >> 3;
            id = ((((kk >> sh) << 3) + (id & 7u)) << sh) + (kk & ((1u << sh) - 1u));
        }
    }
    const size_t g = id;
    if (g >= groups) return;
    const int lane = threadIdx.x, wv = lane >> 6, l = lane & 63;
    const u4v *src = in + (g * 4 + wv) * (size_t)(R * 64) + l;
    u4v r[R];
#pragma unroll
    for (int j = 0; j < R; j++) r[j] = src[j * 64];
    u4v *dst = out + (g * 4 + wv) * (size_t)(W * 64) + l;
    u4v acc = r[0]; // every loaded piece feeds every store: no load can be dropped
#pragma unroll
    for (int j = 1; j < R; j++) acc ^= r[j];
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

static int g_run_kib = 0; // 0: as dispatched; else every XCD takes runs of about this many KiB of the INPUT stream
template <int R, int W>
void run(bool pr) {
    const size_t groups = TOTAL / ((size_t)(R + W) * 4096);
    int sh = -1;
    if (g_run_kib) { // a workgroup reads R * 4 KiB
        sh = 0;
        while ((size_t)(2 << sh) * R * 4 <= (size_t)g_run_kib) sh++;
    }
    CHECK(hipFuncSetAttribute((const void *)k<R, W>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
    if (pr) printf("%2d x 16 B in, %2d x 16 B out per lane:", R, W);
    for (int wg : {8, 6, 4, 3, 2}) {
        const unsigned lds = wg >= 8 ? 0u : (unsigned)(160 * 1024 / wg - 1024) & ~255u;
        const double ms = timeit([&] { hipLaunchKernelGGL((k<R, W>), dim3((unsigned)groups), dim3(256), lds, 0, A, B, groups, sh); }, 15);
        if (pr) printf("  %d wg/CU %5.1f %%", wg, (double)groups * (R + W) * 4096 / (ms * 1e-3) / 8e12 * 100);
    }
    if (pr) printf("   (%d KiB of loads in flight per CU at 8 wg/CU)\n", 32 * R);
}

int main(int argc, char **argv) {
    g_run_kib = argc > 1 ? atoi(argv[1]) : 0;
    if (g_run_kib) printf("every XCD takes runs of about %d KiB of the input stream\n", g_run_kib);
    CHECK(hipMalloc(&A, TOTAL));
    CHECK(hipMalloc(&B, TOTAL));
    CHECK(hipMemset(A, 1, TOTAL));
    CHECK(hipMemset(B, 0, TOTAL));
    for (int pass = 0; pass < 3; pass++) {
        const bool pr = pass > 0;
        if (pr) printf("-- pass %d: %% of 8 TB/s\n", pass);
        run<2, 1>(pr);
        run<4, 2>(pr);
        run<8, 4>(pr);
        run<16, 8>(pr);
        run<1, 2>(pr);
        run<2, 4>(pr);
        run<4, 8>(pr);
        run<8, 16>(pr);
        run<1, 1>(pr);
        run<4, 4>(pr);
        run<8, 8>(pr);
    }
    return 0;
}
