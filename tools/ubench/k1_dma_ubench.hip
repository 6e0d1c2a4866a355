// k1_dma_ubench.hip -- round 4's one bounded attempt at K1's plateau (task 4): decouple the LOAD granule from
// the compute granule.  The shipped k_decode_packed loads a block's 128 B into 32 VGPRs per lane (8 KiB per wave) and is
// at the ceiling of that granule.  Here the coefficients arrive by LDS-DMA (global_load_lds_dwordx4:
// no VGPR destination, 1 KiB contiguous per wave instruction) into a per-wave ring of NBUF x 8 KiB; the wave is
// persistent, keeps NBUF 64-block units in flight, and reads its lane's block back with 8 x ds_read_b128 (XOR swizzle on
// the SOURCE address: conflict-free).  No barrier, no hand-shake: a wave fills and drains its own ring and counts vmcnt.
// Traffic only (plus `work` dependent VALU instructions standing in for the butterfly); the output is a function of the
// loaded bytes that the block-per-lane reference shape also computes, so every variant is checked word for word.
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
struct P {
    const uint4 *coefs;
    unsigned char *pixels;
    size_t coef_fs, pixel_fs; // in uint4 / bytes
    int n_comp, work, tiles, frames, contiguous, run;
    unsigned tiles_magic;
    Comp comp[3];
};

__device__ __forceinline__ unsigned churn(const uint4 (&r)[8], int work) {
    unsigned acc = 0;
    if (work > 0) {
#pragma unroll
        for (int j = 0; j < 8; j++) acc += r[j].x ^ r[j].y ^ r[j].z ^ r[j].w;
        for (int i = 0; i < work; i += 3) acc = (acc ^ (acc << 5)) + 0x9e3779b9u;
    }
    return acc;
}

// the reference shape: k_decode_packed's loads and stores, one block per lane, one tile per workgroup
// p.contiguous (re-used as the XCD mapping of this kernel): 0 = workgroup (x, y) takes tile x of frame y, consecutive tiles on
// consecutive XCDs (the dispatcher deals workgroups round-robin over the 8 XCDs); 1 = every XCD sweeps its own contiguous
// eighth of the batch; 2 = every XCD takes whole frames in turn (frame = 8 * k + xcd)
__global__ __launch_bounds__(256) void k1_ref(P p) {
    extern __shared__ unsigned char dyn_lds[];
    const int lane = threadIdx.x;
    unsigned id = blockIdx.y * gridDim.x + blockIdx.x;
    const unsigned total = gridDim.x * gridDim.y;
    if (p.contiguous == 1) id = (id & 7u) * (total >> 3) + (id >> 3);
    else if (p.contiguous == 2) {
        const unsigned xcd = id & 7u, k = id >> 3, per = gridDim.x; // k-th workgroup of this XCD: tile k % per of its (k / per)-th frame
        id = ((k / per) * 8u + xcd) * per + k % per;
    } else if (p.contiguous == 3) { // runs of p.run consecutive tiles per XCD (p.run divides the total / 8)
        const unsigned xcd = id & 7u, k = id >> 3, per = (unsigned)p.run;
        id = ((k / per) * 8u + xcd) * per + k % per;
    }
    const unsigned frame_y = id / gridDim.x;
    const int tile = (int)(id - frame_y * gridDim.x);
    int c = 0;
#pragma unroll
    for (int i = 1; i < 3; i++)
        if (i < p.n_comp && tile >= p.comp[i].tile0) c = i;
    const Comp &K = p.comp[c];
    int b = (tile - K.tile0) * 256 + lane;
    const bool active = b < K.nblk;
    b = active ? b : K.nblk - 1;
    const unsigned by = __umulhi((unsigned)b, K.magic), bx = (unsigned)b - by * (unsigned)K.bw;
    const uint4 *src = p.coefs + (size_t)frame_y * p.coef_fs + K.coef_off + (size_t)b * 8;
    uint4 r[8];
#pragma unroll
    for (int j = 0; j < 8; j++) r[j] = src[j];
    const unsigned acc = churn(r, p.work);
    unsigned char *dst = p.pixels + (size_t)frame_y * p.pixel_fs + K.plane_off + (size_t)by * 8 * K.stride + (size_t)bx * 8;
    if (active) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const u2v t = {r[j].x ^ r[j].z ^ acc, r[j].y ^ r[j].w};
            __builtin_nontemporal_store(t, reinterpret_cast<u2v *>(dst + (size_t)j * K.stride));
        }
    }
}

template <bool NT>
__device__ __forceinline__ void dma16(const uint4 *gsrc, unsigned lds_dst) {
    unsigned keep;
    if (NT)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// A cursor over the 64-block units a wave owns: unit u = (frame * tiles + tile) * 4 + quarter; units past the end of a
// component's last tile are skipped (so every unit a wave works on has a lane that stores).
struct Cursor {
    int u, step, nunits; // wave-uniform
    int frame, c, block0;
    __device__ __forceinline__ bool settle(const P &p) {
        while (u < nunits) {
            const unsigned item = (unsigned)u >> 2, q = (unsigned)u & 3u;
            frame = (int)__umulhi(item, p.tiles_magic);
            const int tile = (int)item - frame * p.tiles;
            c = 0;
#pragma unroll
            for (int i = 1; i < 3; i++)
                if (i < p.n_comp && tile >= p.comp[i].tile0) c = i;
            block0 = (tile - p.comp[c].tile0) * 256 + (int)q * 64;
            if (block0 < p.comp[c].nblk) return true;
            u += step;
        }
        return false;
    }
};

// NBUF units in flight per wave; WAVES waves per workgroup (they share nothing but the LDS allocation)
template <int NBUF, int WAVES, bool NT>
__global__ __launch_bounds__(WAVES * 64) void k1_dma(P p) {
    extern __shared__ uint4 ring[]; // [WAVES][NBUF][512] pieces of 16 B
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned ring0 = (unsigned)(size_t)(__attribute__((address_space(3))) uint4 *)ring + (unsigned)wave * NBUF * 8192u;
    const int rbase = wave * NBUF * 512;
    // what lane i moves in every DMA instruction k: block k*8 + (i >> 3) of the unit, piece (i & 7) ^ ((i >> 3) & 7)
    const int dblk = lane >> 3, dpiece = (lane & 7) ^ ((lane >> 3) & 7);
    const int nunits = p.tiles * p.frames * 4;
    Cursor fill{(int)blockIdx.x * WAVES + wave, (int)gridDim.x * WAVES, nunits, 0, 0, 0};
    if (p.contiguous) { // every wave walks its own contiguous range of units instead of striding with all the others
        const int nwaves = (int)gridDim.x * WAVES, w = (int)blockIdx.x * WAVES + wave;
        const int per = (nunits + nwaves - 1) / nwaves;
        fill.u = w * per;
        fill.step = 1;
        fill.nunits = (w + 1) * per < nunits ? (w + 1) * per : nunits;
    }
    Cursor work = fill;
    auto issue = [&](const Cursor &cu, int slot) {
        const Comp &K = p.comp[cu.c];
        const uint4 *base = p.coefs + (size_t)cu.frame * p.coef_fs + K.coef_off;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            int blk = cu.block0 + k * 8 + dblk;
            blk = blk < K.nblk ? blk : K.nblk - 1;
            dma16<NT>(base + (size_t)blk * 8 + dpiece, ring0 + (unsigned)slot * 8192u + (unsigned)k * 1024u);
        }
    };
    // prologue: NBUF units on their way
    int filled = 0;
    bool tail = false;
#pragma unroll
    for (int s = 0; s < NBUF; s++) {
        if (fill.settle(p)) {
            issue(fill, s);
            fill.u += fill.step;
            filled++;
        } else
            tail = true;
    }
    int slot = 0, done = 0;
    while (work.settle(p)) {
        // the unit in `slot` was issued NBUF units ago; behind it in the queue: (NBUF - 1) fills and NBUF store groups --
        // once the wave is NBUF units into its walk; before that only the prologue's other fills are sure to be there
        if (tail)
            wait_vm<0>();
        else if (done < NBUF)
            wait_vm<8 * (NBUF - 1)>();
        else
            wait_vm<8 * (2 * NBUF - 1) < 63 ? 8 * (2 * NBUF - 1) : 63>();
        done++;
        const Comp &K = p.comp[work.c];
        int b = work.block0 + lane;
        const bool active = b < K.nblk;
        b = active ? b : K.nblk - 1;
        uint4 r[8];
#pragma unroll
        for (int j = 0; j < 8; j++) r[j] = ring[rbase + slot * 512 + lane * 8 + (j ^ (lane & 7))];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // the slot is free again: refill it before anything else
        if (fill.settle(p)) {
            issue(fill, slot);
            fill.u += fill.step;
        } else
            tail = true;
        const unsigned acc = churn(r, p.work);
        const unsigned by = __umulhi((unsigned)b, K.magic), bx = (unsigned)b - by * (unsigned)K.bw;
        unsigned char *dst = p.pixels + (size_t)work.frame * p.pixel_fs + K.plane_off + (size_t)by * 8 * K.stride + (size_t)bx * 8;
        if (active) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const u2v t = {r[j].x ^ r[j].z ^ acc, r[j].y ^ r[j].w};
                __builtin_nontemporal_store(t, reinterpret_cast<u2v *>(dst + (size_t)j * K.stride));
            }
        }
        work.u += work.step;
        slot = slot + 1 == NBUF ? 0 : slot + 1;
    }
    wait_vm<0>();
}

__global__ void k_fill(uint4 *a, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 0x9e3779b9u + (unsigned)(i >> 32);
        x ^= x >> 15;
        x *= 0x85ebca6bu;
        a[i] = make_uint4(x, x ^ 0x1234567u, x * 3u, x + 77u);
    }
}
__global__ void k_diff(const uint4 *a, const uint4 *b, size_t n, unsigned long long *count) {
    unsigned long long bad = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 x = a[i], y = b[i];
        bad += (x.x != y.x) + (x.y != y.y) + (x.z != y.z) + (x.w != y.w);
    }
    if (bad) atomicAdd(count, bad);
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
    CHECK(hipEventDestroy(e0));
    CHECK(hipEventDestroy(e1));
    return ms / reps;
}

static void geometry(P &p, const uint4 *a, unsigned char *b, size_t &blocks) {
    std::memset(&p, 0, sizeof p);
    p.coefs = a;
    p.pixels = b;
    const int geo[3][2] = {{240, 136}, {120, 68}, {120, 68}}; // 1080p 4:2:0 as the decoder pads it
    p.n_comp = 3;
    size_t co = 0, po = 0;
    int t = 0;
    blocks = 0;
    for (int i = 0; i < 3; i++) {
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
    p.tiles = t;
    p.tiles_magic = (unsigned)(((1ull << 32) + t - 1) / t);
}

struct Variant {
    const char *name;
    const void *fn;
    int threads, lds;
};

template <int NBUF, int WAVES, bool NT>
static Variant variant(const char *name) {
    return Variant{name, (const void *)k1_dma<NBUF, WAVES, NT>, WAVES * 64, NBUF * WAVES * 8192};
}

int main(int argc, char **argv) {
    const int frames = argc > 1 ? atoi(argv[1]) : 1024;
    const int passes = argc > 2 ? atoi(argv[2]) : 3;
    const bool only_ref = argc > 3; // a third argument: the block-per-lane shape's XCD mappings alone
    P p;
    size_t blocks;
    geometry(p, nullptr, nullptr, blocks);
    p.frames = frames;
    const size_t n_in = (size_t)frames * p.coef_fs, n_out = (size_t)frames * p.pixel_fs;
    uint4 *a;
    unsigned char *out_ref, *out;
    unsigned long long *bad;
    CHECK(hipMalloc(&a, n_in * 16));
    CHECK(hipMalloc(&out_ref, n_out));
    CHECK(hipMalloc(&out, n_out));
    CHECK(hipMalloc(&bad, 8));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, a, n_in);
    CHECK(hipMemset(out_ref, 0, n_out));
    CHECK(hipDeviceSynchronize());
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const double bytes = (double)frames * blocks * 192;
    const int reps = 20;
    printf("%d frames of 1080p 4:2:0 (%zu blocks each), %.2f GB per launch, %d CUs\n", frames, blocks, bytes / 1e9, cus);
    auto report = [&](const char *name, double ms, const char *note) {
        printf("%-74s %8.4f ms  %5.1f %% of 8 TB/s%s\n", name, ms, bytes / (ms * 1e-3) / 8e12 * 100, note);
        fflush(stdout);
    };
    const Variant variants[] = {
        variant<1, 4, false>("LDS-DMA, 1 unit in flight per wave, 4 waves per workgroup"),
        variant<2, 4, false>("LDS-DMA, 2 units in flight per wave, 4 waves per workgroup"),
        variant<2, 2, false>("LDS-DMA, 2 units in flight per wave, 2 waves per workgroup"),
        variant<2, 1, false>("LDS-DMA, 2 units in flight per wave, 1 wave per workgroup"),
        variant<3, 1, false>("LDS-DMA, 3 units in flight per wave, 1 wave per workgroup"),
        variant<4, 1, false>("LDS-DMA, 4 units in flight per wave, 1 wave per workgroup"),
        variant<1, 4, true>("LDS-DMA nt, 1 unit in flight per wave, 4 waves per workgroup"),
        variant<2, 2, true>("LDS-DMA nt, 2 units in flight per wave, 2 waves per workgroup"),
        variant<2, 1, true>("LDS-DMA nt, 2 units in flight per wave, 1 wave per workgroup"),
        variant<3, 1, true>("LDS-DMA nt, 3 units in flight per wave, 1 wave per workgroup"),
    };
    for (int pass = 0; pass < passes; pass++) {
        printf("-- pass %d\n", pass);
        for (int work : {0, 400}) for (int contiguous : {0, 1}) {
            P pr = p;
            pr.contiguous = contiguous;
            pr.coefs = a;
            pr.work = work;
            pr.pixels = out_ref;
            char name[200];
            if (!contiguous) for (int wg : {8, 4}) {
                const unsigned lds = wg >= 8 ? 0u : (unsigned)(160 * 1024 / wg - 1024) & ~255u;
                CHECK(hipFuncSetAttribute((const void *)k1_ref, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
                for (int xcd : {2, 1008, 1012, 1016, 1024, 1032, 1064, 1128, 1256, 1512, 0}) { // (0 last: the plain mapping leaves the reference output for the variants below)
                    P px = pr;
                    px.contiguous = xcd >= 1000 ? 3 : xcd;
                    px.run = xcd >= 1000 ? (xcd == 3072 ? 1536 : xcd - 1000) : 0;
                    CHECK(hipMemsetAsync(out_ref, 0, n_out));
                    const double t = timeit([&] { hipLaunchKernelGGL(k1_ref, dim3(p.tiles, frames), dim3(256), lds, 0, px); }, reps);
                    snprintf(name, sizeof name, "block per lane (K1's shape), %d VALU, %d workgroups per CU%s", work, wg,
                             xcd == 1 ? ", every XCD its own eighth of the batch" : xcd == 2 ? ", every XCD whole frames in turn" : "");
                    if (xcd >= 1000) snprintf(name + strlen(name), sizeof name - strlen(name), ", every XCD runs of %d tiles", px.run);
                    report(name, t, "");
                }
            }
            if (!only_ref) for (const Variant &v : variants) {
                P pv = pr;
                pv.pixels = out;
                CHECK(hipFuncSetAttribute(v.fn, hipFuncAttributeMaxDynamicSharedMemorySize, v.lds));
                int per_cu = 0;
                CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, v.fn, v.threads, v.lds));
                if (per_cu < 1) {
                    printf("%s: does not fit\n", v.name);
                    continue;
                }
                void *args[] = {&pv};
                const dim3 grid(cus * per_cu), block(v.threads);
                CHECK(hipMemsetAsync(out, 0, n_out));
                const double t = timeit([&] { CHECK(hipLaunchKernel(v.fn, grid, block, args, v.lds, 0)); }, reps);
                CHECK(hipMemsetAsync(bad, 0, 8));
                hipLaunchKernelGGL(k_diff, dim3(4096), dim3(256), 0, 0, (const uint4 *)out_ref, (const uint4 *)out, n_out / 16, bad);
                unsigned long long h = 0;
                CHECK(hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost));
                char note[96];
                snprintf(note, sizeof note, "  [%d waves per CU%s]", per_cu * v.threads / 64, h ? "; OUTPUT DIFFERS" : "; output equal");
                snprintf(name, sizeof name, "%s, %d VALU%s", v.name, work, contiguous ? ", own range" : "");
                report(name, t, note);
                if (h) printf("   !! %llu words differ from the reference shape's output\n", h);
            }
        }
    }
    return 0;
}
