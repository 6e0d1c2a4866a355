// k1_ubench.hip -- what separates k_decode_packed (72-73 % of 8 TB/s) from the pure-traffic run of its own access shape
// on luma-only planes (75 % in shape_ubench, 76.7 % for the ideal 2:1 mix on the same box)?  One factor at a time:
//   geometry   luma-only planes 240 x 135  |  the real 1080p 4:2:0 frame record: Y 240 x 136, Cb / Cr 120 x 68 (192 tiles)
//   occupancy  workgroups per CU held down by dynamic LDS (the shipped kernel runs 4 waves / SIMD = 4 workgroups / CU)
//   kernarg    a 2 KB parameter block like DecodeParams (tables in the kernarg segment)
//   work       ~N dependent VALU instructions between the loads and the stores (the shipped kernel: ~860 per wave)
//   tail       the fix-up list epilogue (ballot; never taken)
// Pure traffic otherwise: 8 x 16 B loads per lane (the block's 128 B), 8 x 8 B non-temporal row stores.
// Measurement tool only -- not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

typedef unsigned u2v __attribute__((ext_vector_type(2)));

struct Comp {
    int bw, nblk, tile0;
    unsigned magic;
    size_t coef_off, plane_off, stride;
};
struct Params {
    const uint4 *coefs;
    unsigned char *pixels;
    size_t coef_fs, pixel_fs; // in uint4 / bytes
    int n_comp, work;
    Comp comp[3];
    unsigned *fix_count, *fix_list;
    int run_sh; // >= 0: every XCD takes runs of 2^run_sh consecutive tiles (csrc/hvc_kernels.h xcd_work); -1: as dispatched
    int pad[479]; // ~2 KB like DecodeParams (qt, qpair, thresholds)
};
struct ParamsSmall {
    const uint4 *coefs;
    unsigned char *pixels;
    size_t coef_fs, pixel_fs;
    int n_comp, work;
    Comp comp[3];
    unsigned *fix_count, *fix_list;
    int run_sh;
};

// workgroup -> (frame, tile) as csrc/hvc_kernels.h xcd_work does it (round 4)
__device__ __forceinline__ void xcd_remap(int sh, unsigned &frame, unsigned &tile) {
    frame = blockIdx.y;
    tile = blockIdx.x;
    if (sh < 0) return;
    const unsigned per = gridDim.x, id = frame * per + tile, group = 8u << sh, total = per * gridDim.y;
    if (id >= total - total % group) return;
    const unsigned k = id >> 3, lin = ((((k >> sh) << 3) + (id & 7u)) << sh) + (k & ((1u << sh) - 1u));
    frame = lin / per;
    tile = lin - frame * per;
}

template <class P, bool TAIL>
__global__ __launch_bounds__(256) void k1(P p) {
    extern __shared__ unsigned char dyn_lds[]; // occupancy control only
    unsigned wframe, wtile;
    xcd_remap(p.run_sh, wframe, wtile);
    const int lane = threadIdx.x, tile = (int)wtile;
    int c = 0;
#pragma unroll
    for (int i = 1; i < 3; i++)
        if (i < p.n_comp && tile >= p.comp[i].tile0) c = i;
    const Comp &K = p.comp[c];
    int b = (tile - K.tile0) * 256 + lane;
    const bool active = b < K.nblk;
    b = active ? b : K.nblk - 1;
    const unsigned by = __umulhi((unsigned)b, K.magic), bx = (unsigned)b - by * (unsigned)K.bw;
    const uint4 *src = p.coefs + (size_t)wframe * p.coef_fs + K.coef_off + (size_t)b * 8;
    uint4 r[8];
#pragma unroll
    for (int j = 0; j < 8; j++) r[j] = src[j];
    unsigned acc = 0;
    if (p.work > 0) { // a dependent chain that needs every loaded dword first (the guard energy of the real kernel does)
#pragma unroll
        for (int j = 0; j < 8; j++) acc += r[j].x ^ r[j].y ^ r[j].z ^ r[j].w;
        for (int i = 0; i < p.work; i += 3) acc = (acc ^ (acc << 5)) + 0x9e3779b9u; // three VALU instructions a trip
    }
    unsigned char *dst = p.pixels + (size_t)wframe * p.pixel_fs + K.plane_off + (size_t)by * 8 * K.stride + (size_t)bx * 8;
    const bool bad = TAIL && acc == 0x12345u && p.work > 0;
    if (active && !bad) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const u2v t = {r[j].x ^ r[j].z ^ acc, r[j].y ^ r[j].w};
            __builtin_nontemporal_store(t, reinterpret_cast<u2v *>(dst + (size_t)j * K.stride));
        }
    }
    if (TAIL) {
        const bool flag = active && bad;
        const unsigned long long m = __ballot(flag);
        if (m) {
            const int wl = lane & 63;
            unsigned base = 0;
            if (wl == 0) base = atomicAdd(p.fix_count, (unsigned)__popcll(m));
            base = __shfl(base, 0);
            if (flag) p.fix_list[base + (unsigned)__popcll(m & ((1ull << wl) - 1ull))] = (unsigned)b;
        }
    }
}

// TWO lanes per block: both lanes of a pair load the block's 128 B (the same addresses: one request), lane 0 of the pair
// stores rows 0-3, lane 1 rows 4-7 -- a wave moves 4 KiB in + 2 KiB out instead of 8 + 4.  LANES = 2 or 4 (4 lanes per
// block: two rows each).
template <int LANES>
__global__ __launch_bounds__(256) void k1_split(ParamsSmall p) {
    extern __shared__ unsigned char dyn_lds[];
    unsigned wframe, wtile;
    xcd_remap(p.run_sh, wframe, wtile);
    const int lane = threadIdx.x, tile = (int)wtile;
    constexpr int BPW = 256 / LANES; // blocks per workgroup
    int c = 0;
#pragma unroll
    for (int i = 1; i < 3; i++)
        if (i < p.n_comp && tile >= p.comp[i].tile0) c = i;
    const Comp &K = p.comp[c];
    int b = (tile - K.tile0) * BPW + lane / LANES;
    const int part = lane % LANES;
    const bool active = b < K.nblk;
    b = active ? b : K.nblk - 1;
    const unsigned by = __umulhi((unsigned)b, K.magic), bx = (unsigned)b - by * (unsigned)K.bw;
    const uint4 *src = p.coefs + (size_t)wframe * p.coef_fs + K.coef_off + (size_t)b * 8;
    uint4 r[8];
#pragma unroll
    for (int j = 0; j < 8; j++) r[j] = src[j];
    unsigned char *dst = p.pixels + (size_t)wframe * p.pixel_fs + K.plane_off + (size_t)by * 8 * K.stride + (size_t)bx * 8;
    if (active) {
#pragma unroll
        for (int j = 0; j < 8 / LANES; j++) {
            const int row = part * (8 / LANES) + j;
            const u2v t = {r[j].x ^ r[j + 8 / LANES > 7 ? 7 : j + 8 / LANES].z ^ r[(j + 3) & 7].y, r[j].y ^ r[(j + 5) & 7].w ^ r[(j + 6) & 7].x};
            __builtin_nontemporal_store(t, reinterpret_cast<u2v *>(dst + (size_t)row * K.stride));
        }
    }
}

template <class F>
double timeit(F launch, int reps) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 4; i++) launch();
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; i++) launch();
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

template <class P>
static void fill(P &p, bool chroma, const uint4 *a, unsigned char *b, unsigned *fix, int &tiles, size_t &blocks) {
    std::memset(&p, 0, sizeof p);
    p.coefs = a;
    p.pixels = b;
    p.fix_count = fix;
    p.fix_list = fix + 64;
    const int geo[3][2] = {{240, chroma ? 136 : 135}, {120, 68}, {120, 68}};
    p.n_comp = chroma ? 3 : 1;
    size_t co = 0, po = 0;
    int t = 0;
    blocks = 0;
    for (int i = 0; i < p.n_comp; i++) {
        Comp &K = p.comp[i];
        K.bw = geo[i][0];
        K.nblk = geo[i][0] * geo[i][1];
        K.tile0 = t;
        K.magic = (unsigned)(((1ull << 32) + K.bw - 1) / K.bw);
        K.coef_off = co;
        K.plane_off = po;
        K.stride = (size_t)K.bw * 8;
        t += (K.nblk + 255) / 256;
        co += (size_t)K.nblk * 8;
        po += (size_t)K.nblk * 64;
        blocks += K.nblk;
    }
    p.coef_fs = co;
    p.pixel_fs = po;
    tiles = t;
}

int main(int argc, char **argv) {
    // argv[1]: run length in 256-block tiles per XCD (0 = workgroups as dispatched), a power of two; the split kernels'
    // smaller tiles take proportionally longer runs (the same bytes per run)
    const int run = argc > 1 ? atoi(argv[1]) : 0;
    int run_sh = -1;
    if (run > 0) { run_sh = 0; while ((2 << run_sh) <= run) run_sh++; }
    printf("workgroup order: %s\n", run_sh < 0 ? "as dispatched" : "runs per XCD");
    const size_t total = 9600ull * 1000000ull;
    uint4 *a;
    unsigned char *b;
    unsigned *fix;
    CHECK(hipMalloc(&a, total * 2 / 3 + (64 << 20)));
    CHECK(hipMalloc(&b, total / 3 + (64 << 20)));
    CHECK(hipMalloc(&fix, 1 << 20));
    CHECK(hipMemset(a, 1, total * 2 / 3));
    CHECK(hipMemset(b, 0, total / 3));
    CHECK(hipMemset(fix, 0, 1 << 20));
    const int reps = 20;
    auto report = [](const char *name, double ms, double bytes) {
        printf("%-78s %8.4f ms  %5.1f %% of 8 TB/s\n", name, ms, bytes / (ms * 1e-3) / 8e12 * 100);
    };
    for (int pass = 0; pass < 3; pass++) {
        const bool pr = pass > 0;
        if (pr) printf("-- pass %d\n", pass);
        for (int chroma = 0; chroma < 2; chroma++) {
            ParamsSmall ps;
            Params pl;
            int tiles;
            size_t blocks;
            fill(ps, chroma, a, b, fix, tiles, blocks);
            fill(pl, chroma, a, b, fix, tiles, blocks);
            ps.run_sh = pl.run_sh = run_sh;
            const int frames = (int)(total / (blocks * 192));
            const double bytes = (double)frames * blocks * 192;
            const dim3 grid(tiles, frames);
            char name[160];
            double t = timeit([&] { hipLaunchKernelGGL((k1<ParamsSmall, false>), grid, dim3(256), 0, 0, ps); }, reps);
            snprintf(name, sizeof name, "%s, small kernarg, no work, free occupancy", chroma ? "4:2:0 frame record (Y + Cb + Cr)" : "luma-only planes");
            if (pr) report(name, t, bytes);
            if (!chroma) continue;
            t = timeit([&] { hipLaunchKernelGGL((k1<Params, false>), grid, dim3(256), 0, 0, pl); }, reps);
            if (pr) report("  + 2 KB kernarg", t, bytes);
            t = timeit([&] { hipLaunchKernelGGL((k1<Params, true>), grid, dim3(256), 0, 0, pl); }, reps);
            if (pr) report("  + 2 KB kernarg + fix-list tail", t, bytes);
            { // the granule halved / quartered: two / four lanes per block (tiles of 128 / 64 blocks)
                fill(ps, chroma, a, b, fix, tiles, blocks);
                ps.run_sh = run_sh;
                for (int lanes : {2, 4}) {
                    ParamsSmall q2 = ps;
                    q2.run_sh = run_sh < 0 ? -1 : run_sh + (lanes == 2 ? 1 : 2);
                    int t2 = 0;
                    for (int i = 0; i < q2.n_comp; i++) {
                        q2.comp[i].tile0 = t2;
                        t2 += (q2.comp[i].nblk + 256 / lanes - 1) / (256 / lanes);
                    }
                    for (int wg : {8, 4}) {
                        const unsigned lds = wg >= 8 ? 0u : (unsigned)(160 * 1024 / wg - 1024) & ~255u;
                        if (lanes == 2) {
                            CHECK(hipFuncSetAttribute((const void *)k1_split<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
                            t = timeit([&] { hipLaunchKernelGGL((k1_split<2>), dim3(t2, frames), dim3(256), lds, 0, q2); }, reps);
                        } else {
                            CHECK(hipFuncSetAttribute((const void *)k1_split<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
                            t = timeit([&] { hipLaunchKernelGGL((k1_split<4>), dim3(t2, frames), dim3(256), lds, 0, q2); }, reps);
                        }
                        snprintf(name, sizeof name, "  %d lanes per block (each stores %d rows), %d workgroups per CU", lanes, 8 / lanes, wg);
                        if (pr) report(name, t, bytes);
                    }
                }
            }
            for (int wg : {8, 6, 5, 4, 3, 2, 1}) { // workgroups per CU by dynamic LDS (160 KB per CU)
                const unsigned lds = wg >= 8 ? 0u : (unsigned)(160 * 1024 / wg - 1024) & ~255u;
                CHECK(hipFuncSetAttribute((const void *)k1<ParamsSmall, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
                t = timeit([&] { hipLaunchKernelGGL((k1<ParamsSmall, false>), grid, dim3(256), lds, 0, ps); }, reps);
                snprintf(name, sizeof name, "  occupancy: %d workgroups (%d waves / SIMD) per CU", wg, wg);
                if (pr) report(name, t, bytes);
            }
            for (int work : {200, 400, 800, 1600}) {
                Params pw = pl;
                pw.work = work;
                for (int wg : {8, 4}) {
                    const unsigned lds = wg >= 8 ? 0u : (unsigned)(160 * 1024 / wg - 1024) & ~255u;
                    CHECK(hipFuncSetAttribute((const void *)k1<Params, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
                    t = timeit([&] { hipLaunchKernelGGL((k1<Params, true>), grid, dim3(256), lds, 0, pw); }, reps);
                    snprintf(name, sizeof name, "  2 KB kernarg + tail + %d dependent VALU instructions, %d workgroups per CU", work, wg);
                    if (pr) report(name, t, bytes);
                }
            }
        }
    }
    return 0;
}
