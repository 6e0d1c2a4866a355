// valu_ubench.hip -- measures the issue cost of the integer VALU instructions the
// JPEG kernels are built from, on gfx950.  Build: make -C tools/ubench ; run on the
// GPU box: build/valu_ubench  (prints cycles per wave-instruction per SIMD at 1, 2,
// 4 and 8 waves per SIMD).  Measurement tool only -- not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

constexpr int ITERS = 2000;   // loop trips
constexpr int UNROLL = 16;    // instructions per trip (4 independent chains x 4)

// BODY uses registers a,b,c,d (chains) and k0,k1 (operands)
#define KERNEL(NAME, BODY)                                                              \
    __global__ __launch_bounds__(256) void NAME(int *out, long long *cyc, int s0, int s1) { \
        int a = threadIdx.x + s0, b = a * 3 + s1, c = a ^ 0x55, d = a + 7;              \
        int k0 = s0 | 1, k1 = s1 | 3;                                                   \
        long long t0 = __builtin_amdgcn_s_memtime();                                    \
        for (int i = 0; i < ITERS; i++) {                                               \
            BODY BODY BODY BODY                                                         \
        }                                                                               \
        long long t1 = __builtin_amdgcn_s_memtime();                                    \
        out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d;                            \
        if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0; \
    }

#define I4(INS) asm volatile(INS(a) "\n" INS(b) "\n" INS(c) "\n" INS(d) : [a] "+v"(a), [b] "+v"(b), [c] "+v"(c), [d] "+v"(d) : [k0] "v"(k0), [k1] "v"(k1));

#define R(r) "%[" #r "]"
#define K0 "%[k0]"
#define K1 "%[k1]"
#define OP_ADD(r)     "v_add_u32 " R(r) ", " R(r) ", " K0
#define OP_SUB(r)     "v_sub_u32 " R(r) ", " R(r) ", " K0
#define OP_MUL24(r)   "v_mul_i32_i24 " R(r) ", " R(r) ", " K0
#define OP_MAD24(r)   "v_mad_i32_i24 " R(r) ", " R(r) ", " K0 ", " K1
#define OP_MADU24(r)  "v_mad_u32_u24 " R(r) ", " R(r) ", " K0 ", " K1
#define OP_MADI16(r)  "v_mad_i32_i16 " R(r) ", " R(r) ", " K0 ", " K1
#define OP_MULLO(r)   "v_mul_lo_u32 " R(r) ", " R(r) ", " K0
#define OP_MULHI(r)   "v_mul_hi_u32 " R(r) ", " R(r) ", " K0
#define OP_MAX3(r)    "v_max3_i32 " R(r) ", " R(r) ", " K0 ", " K1
#define OP_MED3(r)    "v_med3_i32 " R(r) ", " R(r) ", " K0 ", " K1
#define OP_ADD3(r)    "v_add3_u32 " R(r) ", " R(r) ", " K0 ", " K1
#define OP_OR3(r)     "v_or3_b32 " R(r) ", " R(r) ", " K0 ", " K1
#define OP_XAD(r)     "v_xad_u32 " R(r) ", " R(r) ", " K0 ", " K1
#define OP_LSHLADD(r) "v_lshl_add_u32 " R(r) ", " R(r) ", 3, " K1
#define OP_ASHR(r)    "v_ashrrev_i32 " R(r) ", 1, " R(r)
#define OP_DOT2(r)    "v_dot2_i32_i16 " R(r) ", " K0 ", " K1 ", " R(r)
#define OP_ASHRPK(r)  "v_ashr_pk_u8_i32 " R(r) ", " R(r) ", " K0 ", 3"
#define OP_SDWA(r)    "v_mul_i32_i24_sdwa " R(r) ", sext(" R(r) "), " K0 " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD"
#define OP_CVTF(r)    "v_cvt_f32_i32 " R(r) ", " R(r)
#define OP_FMA(r)     "v_fma_f32 " R(r) ", " R(r) ", " K0 ", " K1
#define OP_PERM(r)    "v_perm_b32 " R(r) ", " R(r) ", " K0 ", " K1
#define OP_BFI(r)     "v_bfi_b32 " R(r) ", " K0 ", " R(r) ", " K1
#define OP_CVTPK(r)   "v_cvt_pk_i16_i32 " R(r) ", " R(r) ", " K0
#define OP_PKADD(r)   "v_pk_add_i16 " R(r) ", " R(r) ", " K0

#define OP_LSHL(r)    "v_lshlrev_b32 " R(r) ", 3, " R(r)
#define OP_AND(r)     "v_and_b32 " R(r) ", " R(r) ", " K0
#define OP_OR(r)      "v_or_b32 " R(r) ", " R(r) ", " K0
#define OP_XOR(r)     "v_xor_b32 " R(r) ", " R(r) ", " K0
#define OP_MAXI(r)    "v_max_i32 " R(r) ", " R(r) ", " K0
#define OP_MAXU(r)    "v_max_u32 " R(r) ", " R(r) ", " K0
#define OP_MOV(r)     "v_mov_b32 " R(r) ", " K0
#define OP_BFE(r)     "v_bfe_i32 " R(r) ", " R(r) ", 0, 16"
#define OP_LSHLOR(r)  "v_lshl_or_b32 " R(r) ", " R(r) ", 8, " K1
#define OP_ANDOR(r)   "v_and_or_b32 " R(r) ", " R(r) ", " K0 ", " K1
#define OP_PKMUL(r)   "v_pk_mul_lo_u16 " R(r) ", " R(r) ", " K0
#define OP_PKMAD(r)   "v_pk_mad_i16 " R(r) ", " R(r) ", " K0 ", " K1
#define OP_SATPK(r)   "v_sat_pk_u8_i16 " R(r) ", " R(r)
#define OP_CVTI(r)    "v_cvt_i32_f32 " R(r) ", " R(r)
#define OP_FLOOR(r)   "v_floor_f32 " R(r) ", " R(r)
#define OP_MULF(r)    "v_mul_f32 " R(r) ", " R(r) ", " K0
#define OP_ADDF(r)    "v_add_f32 " R(r) ", " R(r) ", " K0
#define OP_MAXF(r)    "v_max_f32 " R(r) ", " R(r) ", " K0
#define OP_MAXFA(r)   "v_max_f32 " R(r) ", |" R(r) "|, |" K0 "|"
#define OP_MAX3F(r)   "v_max3_f32 " R(r) ", " R(r) ", " K0 ", " K1
#define OP_FMAC(r)    "v_fmac_f32 " R(r) ", " K0 ", " K1
#define OP_ADDLIT(r)  "v_add_u32 " R(r) ", 0x12345, " R(r)
#define OP_MULLIT(r)  "v_mul_i32_i24 " R(r) ", 0xb19, " R(r)
#define OP_SUBSD(r)   "v_sub_u32_sdwa " R(r) ", " R(r) ", " K0 " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD"
#define OP_ASHRSD(r)  "v_ashrrev_i32_sdwa " R(r) ", " K0 ", sext(" R(r) ") dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0"
#define OP_ALIGN(r)   "v_alignbit_b32 " R(r) ", " R(r) ", " K0 ", 8"
#define OP_CNDM(r)    "v_cndmask_b32 " R(r) ", " R(r) ", " K0 ", vcc"
#define OP_DPPMOV(r)  "v_mov_b32_dpp " R(r) ", " R(r) " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
#define OP_ADDDPP(r)  "v_add_u32_dpp " R(r) ", " R(r) ", " K0 " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"

KERNEL(k_add, I4(OP_ADD))
KERNEL(k_sub, I4(OP_SUB))
KERNEL(k_mul24, I4(OP_MUL24))
KERNEL(k_mad24, I4(OP_MAD24))
KERNEL(k_madu24, I4(OP_MADU24))
KERNEL(k_mullo, I4(OP_MULLO))
KERNEL(k_mulhi, I4(OP_MULHI))
KERNEL(k_max3, I4(OP_MAX3))
KERNEL(k_med3, I4(OP_MED3))
KERNEL(k_add3, I4(OP_ADD3))
KERNEL(k_lshladd, I4(OP_LSHLADD))
KERNEL(k_ashr, I4(OP_ASHR))
KERNEL(k_dot2, I4(OP_DOT2))
KERNEL(k_ashrpk, I4(OP_ASHRPK))
KERNEL(k_sdwa, I4(OP_SDWA))
KERNEL(k_cvtf, I4(OP_CVTF))
KERNEL(k_fma, I4(OP_FMA))
KERNEL(k_perm, I4(OP_PERM))
KERNEL(k_bfi, I4(OP_BFI))
KERNEL(k_cvtpk, I4(OP_CVTPK))
KERNEL(k_pkadd, I4(OP_PKADD))
KERNEL(k_or3, I4(OP_OR3))
KERNEL(k_xad, I4(OP_XAD))
KERNEL(k_madi16, I4(OP_MADI16))

KERNEL(k_lshl, I4(OP_LSHL))
KERNEL(k_and, I4(OP_AND))
KERNEL(k_or, I4(OP_OR))
KERNEL(k_xor, I4(OP_XOR))
KERNEL(k_maxi, I4(OP_MAXI))
KERNEL(k_maxu, I4(OP_MAXU))
KERNEL(k_mov, I4(OP_MOV))
KERNEL(k_bfe, I4(OP_BFE))
KERNEL(k_lshlor, I4(OP_LSHLOR))
KERNEL(k_andor, I4(OP_ANDOR))
KERNEL(k_pkmul, I4(OP_PKMUL))
KERNEL(k_pkmad, I4(OP_PKMAD))
KERNEL(k_satpk, I4(OP_SATPK))
KERNEL(k_cvti, I4(OP_CVTI))
KERNEL(k_floor, I4(OP_FLOOR))
KERNEL(k_mulf, I4(OP_MULF))
KERNEL(k_addf, I4(OP_ADDF))
KERNEL(k_maxf, I4(OP_MAXF))
KERNEL(k_maxfa, I4(OP_MAXFA))
KERNEL(k_max3f, I4(OP_MAX3F))
KERNEL(k_fmac, I4(OP_FMAC))
KERNEL(k_addlit, I4(OP_ADDLIT))
KERNEL(k_mullit, I4(OP_MULLIT))
KERNEL(k_subsd, I4(OP_SUBSD))
KERNEL(k_ashrsd, I4(OP_ASHRSD))
KERNEL(k_align, I4(OP_ALIGN))
KERNEL(k_cndm, I4(OP_CNDM))
KERNEL(k_dppmov, I4(OP_DPPMOV))
KERNEL(k_adddpp, I4(OP_ADDDPP))

typedef void (*kfn)(int *, long long *, int, int);
struct Entry { const char *name; kfn fn; };

int main() {
    int dev = 0;
    CHECK(hipSetDevice(dev));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, dev));
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, clock %d kHz\n", prop.gcnArchName, cus, prop.clockRate);
    int *out;
    long long *cyc;
    const int max_blocks = cus * 8;
    CHECK(hipMalloc(&out, (size_t)max_blocks * 256 * sizeof(int)));
    CHECK(hipMalloc(&cyc, (size_t)max_blocks * 4 * sizeof(long long)));
    std::vector<long long> h((size_t)max_blocks * 4);
    Entry es[] = {{"v_add_u32", k_add}, {"v_sub_u32", k_sub}, {"v_mul_i32_i24", k_mul24}, {"v_mad_i32_i24", k_mad24},
                  {"v_mad_u32_u24", k_madu24}, {"v_mul_lo_u32", k_mullo}, {"v_mul_hi_u32", k_mulhi}, {"v_max3_i32", k_max3},
                  {"v_med3_i32", k_med3}, {"v_add3_u32", k_add3}, {"v_lshl_add_u32", k_lshladd},
                  {"v_ashrrev_i32", k_ashr}, {"v_dot2_i32_i16", k_dot2}, {"v_ashr_pk_u8_i32", k_ashrpk},
                  {"v_mul_i32_i24_sdwa", k_sdwa}, {"v_cvt_f32_i32", k_cvtf}, {"v_fma_f32", k_fma},
                  {"v_perm_b32", k_perm}, {"v_bfi_b32", k_bfi}, {"v_cvt_pk_i16_i32", k_cvtpk},
                  {"v_pk_add_i16", k_pkadd}, {"v_or3_b32", k_or3}, {"v_xad_u32", k_xad}, {"v_mad_i32_i16", k_madi16}, {"v_lshlrev_b32", k_lshl}, {"v_and_b32", k_and}, {"v_or_b32", k_or}, {"v_xor_b32", k_xor}, {"v_max_i32", k_maxi}, {"v_max_u32", k_maxu}, {"v_mov_b32", k_mov}, {"v_bfe_i32", k_bfe}, {"v_lshl_or_b32", k_lshlor}, {"v_and_or_b32", k_andor}, {"v_pk_mul_lo_u16", k_pkmul}, {"v_pk_mad_i16", k_pkmad}, {"v_sat_pk_u8_i16", k_satpk}, {"v_cvt_i32_f32", k_cvti}, {"v_floor_f32", k_floor}, {"v_mul_f32", k_mulf}, {"v_add_f32", k_addf}, {"v_max_f32", k_maxf}, {"v_max_f32 |abs|", k_maxfa}, {"v_max3_f32", k_max3f}, {"v_fmac_f32", k_fmac}, {"v_add_u32 literal", k_addlit}, {"v_mul_i32_i24 literal", k_mullit}, {"v_sub_u32_sdwa", k_subsd}, {"v_ashrrev_i32_sdwa", k_ashrsd}, {"v_alignbit_b32", k_align}, {"v_cndmask_b32", k_cndm}, {"v_mov_b32_dpp", k_dppmov}, {"v_add_u32_dpp", k_adddpp}};
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    printf("%-22s %10s %10s %10s %10s   (cycles per wave-instruction per SIMD, by waves/SIMD; s_memtime)\n", "instruction",
           "1w", "2w", "4w", "8w");
    for (auto &e : es) {
        printf("%-22s", e.name);
        for (int k : {1, 2, 4, 8}) {
            const int blocks = cus * k; // k blocks of 4 waves per CU -> k waves per SIMD
            hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, out, cyc, 1, 2); // warm
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, out, cyc, 1, 2);
            CHECK(hipEventRecord(e1));
            CHECK(hipDeviceSynchronize());
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            CHECK(hipMemcpy(h.data(), cyc, (size_t)blocks * 4 * sizeof(long long), hipMemcpyDeviceToHost));
            double sum = 0;
            for (int i = 0; i < blocks * 4; i++) sum += (double)h[i];
            double per_wave = sum / (blocks * 4);
            double n_instr = (double)ITERS * UNROLL;
            // s_memtime counts at a fixed 100 MHz on gfx9?  Report both the raw tick ratio and wall-clock.
            double ns_per_instr_per_simd = (double)ms * 1e6 / (n_instr * k);
            printf(" %5.2f/%4.2fns", per_wave / (n_instr * k), ns_per_instr_per_simd);
        }
        printf("\n");
    }
    return 0;
}
