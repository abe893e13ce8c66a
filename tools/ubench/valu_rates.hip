// Issue rate of the VALU instructions the HBM-side kernels of the emulation are made of (quantise, CRT, GEMM epilogue) on gfx950:
// cycles per wave64 instruction on one SIMD, from loops of 16 independent instructions (two waves per SIMD to cover dependent latency).
// The CRT and quantise kernels turned out to be bound by VALU issue, not by HBM (profiles/archive/r03_hbm_ab.txt): this table prices their
// instruction mixes.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

#define REP16(S) S S S S S S S S S S S S S S S S
// ASM: instruction text using %0 (64- or 32-bit destination/accumulator) and %1, %2 (sources)
#define KERNEL64(NAME, ASM)                                                                                     \
    __global__ void __launch_bounds__(512) NAME(double* out, int iters, double s1, double s2) {                  \
        double a[16];                                                                                           \
        for (int i = 0; i < 16; ++i) a[i] = s1 * (threadIdx.x + i);                                             \
        for (int it = 0; it < iters; ++it) {                                                                    \
            _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile(ASM : "+v"(a[i]) : "v"(s1), "v"(s2));  \
        }                                                                                                       \
        double r = 0;                                                                                           \
        for (int i = 0; i < 16; ++i) r += a[i];                                                                 \
        if (r == 1.2345) out[0] = r;                                                                            \
    }
#define KERNEL32(NAME, ASM)                                                                                     \
    __global__ void __launch_bounds__(512) NAME(double* out, int iters, double s1d, double s2d) {                \
        unsigned a[16];                                                                                         \
        const unsigned s1 = (unsigned)s1d + threadIdx.x, s2 = (unsigned)s2d;                                    \
        for (int i = 0; i < 16; ++i) a[i] = s1 * (i + 1);                                                       \
        for (int it = 0; it < iters; ++it) {                                                                    \
            _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile(ASM : "+v"(a[i]) : "v"(s1), "v"(s2));  \
        }                                                                                                       \
        unsigned r = 0;                                                                                         \
        for (int i = 0; i < 16; ++i) r ^= a[i];                                                                 \
        if (r == 0x12345u) out[0] = r;                                                                          \
    }
// 32-bit source -> 64-bit destination
#define KERNEL3264(NAME, ASM)                                                                                   \
    __global__ void __launch_bounds__(512) NAME(double* out, int iters, double s1d, double s2d) {                \
        double a[16];                                                                                           \
        unsigned s[16];                                                                                         \
        for (int i = 0; i < 16; ++i) a[i] = 0, s[i] = (unsigned)s1d * (i + 3) + threadIdx.x;                    \
        for (int it = 0; it < iters; ++it) {                                                                    \
            _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile(ASM : "+v"(a[i]) : "v"(s[i]), "v"(s[i]));  \
        }                                                                                                       \
        double r = 0;                                                                                           \
        for (int i = 0; i < 16; ++i) r += a[i];                                                                 \
        if (r == 1.2345) out[0] = r;                                                                            \
    }

KERNEL64(k_fma_f64, "v_fma_f64 %0, %1, %2, %0")
KERNEL64(k_add_f64, "v_add_f64 %0, %0, %1")
KERNEL64(k_mul_f64, "v_mul_f64 %0, %0, %1")
KERNEL64(k_rndne_f64, "v_rndne_f64 %0, %0")
KERNEL64(k_ldexp_f64, "v_ldexp_f64 %0, %0, 3")
KERNEL64(k_lshr_b64, "v_lshrrev_b64 %0, 3, %0")
KERNEL64(k_lshl_b64, "v_lshlrev_b64 %0, 1, %0")
KERNEL3264(k_cvt_f64_i32, "v_cvt_f64_i32 %0, %1")
KERNEL3264(k_cvt_f64_u32, "v_cvt_f64_u32 %0, %1")
KERNEL3264(k_cvt_f64_f32, "v_cvt_f64_f32 %0, %1")
KERNEL32(k_fma_f32, "v_fma_f32 %0, %1, %2, %0")
KERNEL32(k_add_u32, "v_add_u32 %0, %0, %1")
KERNEL32(k_bfe_i32, "v_bfe_i32 %0, %0, 8, 8")
KERNEL32(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL32(k_dot4, "v_dot4_u32_u8 %0, %1, %2, %0")
KERNEL32(k_mad_i24, "v_mad_i32_i24 %0, %1, %2, %0")
KERNEL32(k_mul_lo, "v_mul_lo_u32 %0, %0, %1")
KERNEL32(k_mul_hi, "v_mul_hi_u32 %0, %0, %1")
KERNEL32(k_cvt_f32_u32, "v_cvt_f32_u32 %0, %0")
KERNEL32(k_cvt_f32_i32, "v_cvt_f32_i32 %0, %0")
KERNEL32(k_cvt_f32_ubyte1, "v_cvt_f32_ubyte1 %0, %1")
KERNEL32(k_cvt_f32_i32_sdwa, "v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1")
KERNEL32(k_perm, "v_perm_b32 %0, %0, %1, %2")
KERNEL32(k_mov_dpp, "v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
KERNEL32(k_lshl_add, "v_lshl_add_u32 %0, %0, 3, %1")
KERNEL32(k_log_f32, "v_log_f32 %0, %0")

typedef void (*Kern)(double*, int, double, double);
int main() {
    double* out;
    CK(hipMalloc(&out, 8));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct T { const char* name; Kern k; };
    const T tests[] = {{"v_fma_f64", k_fma_f64}, {"v_add_f64", k_add_f64}, {"v_mul_f64", k_mul_f64}, {"v_rndne_f64", k_rndne_f64},
                       {"v_ldexp_f64", k_ldexp_f64}, {"v_lshrrev_b64", k_lshr_b64}, {"v_lshlrev_b64", k_lshl_b64},
                       {"v_cvt_f64_i32", k_cvt_f64_i32}, {"v_cvt_f64_u32", k_cvt_f64_u32}, {"v_cvt_f64_f32", k_cvt_f64_f32},
                       {"v_fma_f32", k_fma_f32}, {"v_add_u32", k_add_u32}, {"v_bfe_i32", k_bfe_i32}, {"v_cndmask_b32", k_cndmask},
                       {"v_dot4_u32_u8", k_dot4}, {"v_mad_i32_i24", k_mad_i24}, {"v_mul_lo_u32", k_mul_lo}, {"v_mul_hi_u32", k_mul_hi},
                       {"v_cvt_f32_u32", k_cvt_f32_u32}, {"v_cvt_f32_i32", k_cvt_f32_i32}, {"v_cvt_f32_ubyte1", k_cvt_f32_ubyte1},
                       {"v_cvt_f32_i32_sdwa(byte1,sext)", k_cvt_f32_i32_sdwa}, {"v_perm_b32", k_perm}, {"v_mov_b32_dpp(quad_perm)", k_mov_dpp},
                       {"v_lshl_add_u32", k_lshl_add}, {"v_log_f32", k_log_f32}};
    const int iters = 4096;
    // 512 threads = 8 waves per workgroup = 2 per SIMD; B workgroups per CU -> 2 B waves per SIMD (1, 2, 4: 2 / 4 / 8 waves per SIMD)
    printf("%-34s", "ns per wave-instruction per SIMD at");
    for (int B : {1, 2, 4}) printf("   %d waves/SIMD", 2 * B);
    printf("\n");
    for (const T& t : tests) {
        printf("%-34s", t.name);
        for (int B : {1, 2, 4}) {
            for (int w = 0; w < 2; ++w) t.k<<<cus * B, 512>>>(out, iters, 1.0000001, 3.0);
            hipEventRecord(e0);
            t.k<<<cus * B, 512>>>(out, iters, 1.0000001, 3.0);
            hipEventRecord(e1);
            CK(hipEventSynchronize(e1));
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            printf("   %12.3f", ms * 1e6 / (2.0 * B * iters * 16));
        }
        printf("\n");
    }
    return 0;
}
