// mem_ubench3.hip -- the machine's streaming ceiling PER READ:WRITE MIX (pure traffic, ideal shape: every wave
// instruction moves 1 KiB contiguous, 16 B per lane; no arithmetic), so that each block-stage kernel can be held
// against the ceiling of ITS mix rather than against one copy figure:
//     K1 decode   128 B read : 64 B written    2:1
//     fused 4:4:4 ~1 : 1   (6.27 MB of coefficients read, 6.22 MB of 4:4:4 frame written per 1080p frame)
//     K3 encode   64 B read : 128 B written    1:2
//     K2 upsample 1 B read : 4 B written       1:4
// plus pure reads and pure writes (plain and non-temporal).  ~9.6 GB per launch like bench.py.
// Measurement tool only -- not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

typedef unsigned u4v __attribute__((ext_vector_type(4)));

// thread t of the grid reads R pieces in[j * n + t] and writes W pieces out[j * n + t] (n = threads in the grid)
// g_sh >= 0: the workgroups are not taken in dispatch order (consecutive 4 KiB pieces behind eight different XCDs' L2s) but
// every XCD takes runs of 2^g_sh consecutive pieces (csrc/hvc_kernels.h xcd_work, round 4)
__device__ __forceinline__ unsigned piece_of(int sh) {
    unsigned id = blockIdx.x;
    if (sh >= 0) {
        const unsigned group = 8u << sh, total = gridDim.x;
        if (id < total - total % group) {
            const unsigned k = id >> 3;
            id = ((((k >> sh) << 3) + (id & 7u)) << sh) + (k & ((1u << sh) - 1u));
        }
    }
    return id;
}
template <int R, int W, bool NT>
__global__ __launch_bounds__(256) void mix(const u4v *__restrict__ in, u4v *__restrict__ out, size_t n, unsigned *sink, int sh) {
    const size_t t = (size_t)piece_of(sh) * 256 + threadIdx.x;
    if (t >= n) return;
    u4v acc = {1u, 2u, 3u, 4u};
#pragma unroll
    for (int j = 0; j < R; j++) {
        const u4v v = in[(size_t)j * n + t];
        acc ^= v;
    }
#pragma unroll
    for (int j = 0; j < W; j++) {
        u4v o = acc;
        o.x += (unsigned)j;
        if (NT) __builtin_nontemporal_store(o, out + (size_t)j * n + t);
        else out[(size_t)j * n + t] = o;
    }
    if (W == 0 && acc.x == 0x12345678u && acc.y == 0x9abcdef0u) *sink = acc.z; // keeps the loads alive
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

static bool g_print = false;
static int g_sh = -1;
template <int R, int W, bool NT>
void run(const char *name, const u4v *in, u4v *out, size_t total_bytes, unsigned *sink) {
    const size_t n = total_bytes / 16 / (R + W);
    const double ms = timeit([&] { hipLaunchKernelGGL((mix<R, W, NT>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, in, out, n, sink, g_sh); }, 20);
    if (!g_print) return; // first pass: warm-up
    printf("%-44s %2d:%-2d %8.4f ms %8.1f GB/s  (%.1f %% of 8 TB/s)\n", name, R, W, ms, (double)n * 16 * (R + W) / (ms * 1e-3) / 1e9,
           (double)n * 16 * (R + W) / (ms * 1e-3) / 8e12 * 100);
}

int main(int argc, char **argv) {
    const size_t total = (argc > 1 ? (size_t)atof(argv[1]) : 9600) * 1000000ull; // bytes moved per launch
    u4v *in, *out;
    unsigned *sink;
    CHECK(hipMalloc(&in, total));
    CHECK(hipMalloc(&out, total));
    CHECK(hipMalloc(&sink, 4));
    CHECK(hipMemset(in, 1, total));
    CHECK(hipMemset(out, 0, total));
    for (int rep = 0; rep < 6; rep++) {
        g_print = rep >= 1;
        const int shs[6] = {-1, -1, 4, 6, 7, 9}; // as dispatched (twice: the first pass warms up), then runs of 16 / 64 / 128 / 512 pieces of 4 KiB
        g_sh = shs[rep];
        if (rep == 1) printf("bytes per launch %.1f GB\n", total / 1e9);
        if (rep >= 2) printf("-- every XCD takes runs of %d consecutive 4 KiB pieces of each stream\n", 1 << g_sh);
        run<1, 0, false>("pure read", in, out, total, sink);
        run<0, 1, false>("pure write, plain stores", in, out, total, sink);
        run<0, 1, true>("pure write, nt stores", in, out, total, sink);
        run<2, 1, false>("K1 mix, plain stores", in, out, total, sink);
        run<2, 1, true>("K1 mix, nt stores", in, out, total, sink);
        run<1, 1, false>("copy / fused 4:4:4 mix, plain stores", in, out, total, sink);
        run<1, 1, true>("copy / fused 4:4:4 mix, nt stores", in, out, total, sink);
        run<1, 2, false>("K3 mix, plain stores", in, out, total, sink);
        run<1, 2, true>("K3 mix, nt stores", in, out, total, sink);
        run<1, 4, false>("K2 mix, plain stores", in, out, total, sink);
        run<1, 4, true>("K2 mix, nt stores", in, out, total, sink);
    }
    return 0;
}
