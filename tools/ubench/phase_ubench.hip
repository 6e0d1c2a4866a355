// phase_ubench.hip -- WHY does the streaming ceiling fall with the bytes a lane moves (inflight_ubench)?  Hypothesis:
// long waves fall into step -- a generation of waves loads together, then stores together -- and the memory sees
// alternating read and write phases.  Variants of the 2:1 mix at K1's granule (8 x 16 B in, 4 x 16 B out per lane):
//   base      all loads, then all stores                                   (inflight_ubench's <8,4>)
//   stagger   the same behind a start delay that differs from workgroup to workgroup (s_sleep)
//   halves    load 4, store 2, load 4, store 2: the wave lives as long but each phase is half the size
//   quarters  load 2, store 1, four times
//   pipe      software pipeline over TWO granules per lane: load A; load B, store A; store B  (loads of the next granule
//             in flight while this one is stored -- what a block kernel could do with two blocks per lane)
// and the 1:2 mix at K3's granule (4 in, 8 out) the same way.  Measurement tool only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef unsigned u4v __attribute__((ext_vector_type(4)));

// a workgroup's granule: R KiB read per wave at in + (g * 4 + wave) * R * 64 pieces, W KiB written likewise
template <int R, int W, int PARTS>
__device__ __forceinline__ void granule(const u4v *__restrict__ in, u4v *__restrict__ out, size_t g, int wv, int l) {
    const u4v *src = in + (g * 4 + wv) * (size_t)(R * 64) + l;
    u4v *dst = out + (g * 4 + wv) * (size_t)(W * 64) + l;
#pragma unroll
    for (int p = 0; p < PARTS; p++) {
        u4v r[R / PARTS];
#pragma unroll
        for (int j = 0; j < R / PARTS; j++) r[j] = src[(p * (R / PARTS) + j) * 64];
        u4v acc = r[0];
#pragma unroll
        for (int j = 1; j < R / PARTS; j++) acc ^= r[j];
#pragma unroll
        for (int j = 0; j < W / PARTS; j++) {
            u4v t = acc;
            t.x += (unsigned)j;
            __builtin_nontemporal_store(t, dst + (p * (W / PARTS) + j) * 64);
        }
    }
}

template <int R, int W, int PARTS, bool STAGGER>
__global__ __launch_bounds__(256) void k(const u4v *__restrict__ in, u4v *__restrict__ out, size_t groups) {
    const size_t g = blockIdx.x;
    if (g >= groups) return;
    if (STAGGER) { // 0 .. 63 x 64 cycles, different for neighbouring workgroups
        const unsigned d = (blockIdx.x * 37u) & 63u;
        for (unsigned i = 0; i < d; i++) __builtin_amdgcn_s_sleep(1);
    }
    granule<R, W, PARTS>(in, out, g, threadIdx.x >> 6, threadIdx.x & 63);
}

// two granules per workgroup, software-pipelined: loads of the second in flight while the first is stored
template <int R, int W>
__global__ __launch_bounds__(256) void kpipe(const u4v *__restrict__ in, u4v *__restrict__ out, size_t groups) {
    const size_t g = (size_t)blockIdx.x * 2;
    if (g + 1 >= groups) return;
    const int wv = threadIdx.x >> 6, l = threadIdx.x & 63;
    const u4v *sa = in + (g * 4 + wv) * (size_t)(R * 64) + l, *sb = in + ((g + 1) * 4 + wv) * (size_t)(R * 64) + l;
    u4v *da = out + (g * 4 + wv) * (size_t)(W * 64) + l, *db = out + ((g + 1) * 4 + wv) * (size_t)(W * 64) + l;
    u4v ra[R], rb[R];
#pragma unroll
    for (int j = 0; j < R; j++) ra[j] = sa[j * 64];
#pragma unroll
    for (int j = 0; j < R; j++) rb[j] = sb[j * 64]; // issued before A's data is needed
    u4v acc = ra[0];
#pragma unroll
    for (int j = 1; j < R; j++) acc ^= ra[j];
#pragma unroll
    for (int j = 0; j < W; j++) {
        u4v t = acc;
        t.x += (unsigned)j;
        __builtin_nontemporal_store(t, da + j * 64);
    }
    acc = rb[0];
#pragma unroll
    for (int j = 1; j < R; j++) acc ^= rb[j];
#pragma unroll
    for (int j = 0; j < W; j++) {
        u4v t = acc;
        t.x += (unsigned)j;
        __builtin_nontemporal_store(t, db + j * 64);
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
static double pct(size_t groups, int r, int w, double ms) { return (double)groups * (r + w) * 4096 / (ms * 1e-3) / 8e12 * 100; }

template <int R, int W>
void family(bool pr, const char *name) {
    const size_t groups = TOTAL / ((size_t)(R + W) * 4096) & ~(size_t)1;
    const dim3 grid((unsigned)groups);
    double b = timeit([&] { hipLaunchKernelGGL((k<R, W, 1, false>), grid, dim3(256), 0, 0, A, B, groups); }, 15);
    double s = timeit([&] { hipLaunchKernelGGL((k<R, W, 1, true>), grid, dim3(256), 0, 0, A, B, groups); }, 15);
    double h = timeit([&] { hipLaunchKernelGGL((k<R, W, 2, false>), grid, dim3(256), 0, 0, A, B, groups); }, 15);
    double q = timeit([&] { hipLaunchKernelGGL((k<R, W, 4, false>), grid, dim3(256), 0, 0, A, B, groups); }, 15);
    double p = timeit([&] { hipLaunchKernelGGL((kpipe<R, W>), dim3((unsigned)(groups / 2)), dim3(256), 0, 0, A, B, groups); }, 15);
    if (pr) printf("%-34s base %5.1f   stagger %5.1f   halves %5.1f   quarters %5.1f   two granules pipelined %5.1f   (%% of 8 TB/s)\n", name,
                   pct(groups, R, W, b), pct(groups, R, W, s), pct(groups, R, W, h), pct(groups, R, W, q), pct(groups, R, W, p));
}

int main() {
    CHECK(hipMalloc(&A, TOTAL));
    CHECK(hipMalloc(&B, TOTAL));
    CHECK(hipMemset(A, 1, TOTAL));
    CHECK(hipMemset(B, 0, TOTAL));
    for (int pass = 0; pass < 3; pass++) {
        const bool pr = pass > 0;
        if (pr) printf("-- pass %d\n", pass);
        family<8, 4>(pr, "2:1, 128 B in + 64 B out (K1)");
        family<4, 8>(pr, "1:2, 64 B in + 128 B out (K3)");
        family<8, 8>(pr, "1:1, 128 B in + 128 B out (fused)");
        family<4, 4>(pr, "1:1, 64 B in + 64 B out");
    }
    return 0;
}
