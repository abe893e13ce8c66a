// Pieces of the CRT accumulation shared by the stand-alone kernels (oz2_crt.hip) and the CRT tail of the tile-stationary INT8
// GEMM kernel (oz2_gemm_i8.hip, FUSE != 0): argument block, the mod-P reduction and the typed helpers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace oz2 {

struct CrtArgs {
    const void* Cmid;
    size_t ld_mid;        // elements (int8 or char2 or int16...) between columns
    size_t plane_stride;  // elements between residue planes
    size_t m, n;
    const int16_t* sftA;
    const int16_t* sftB;
    void* C;
    size_t ldc;
    unsigned N;
    int use_dd;
    int mode;  // 0 general(host scalars) 1: C=AB 2: C+=AB 3: C=-AB 4: C-=AB 5: general(device scalars)
    double alpha[2], beta[2];
    const void* alpha_dev;
    const void* beta_dev;
    double Phi, Plo, invP;
    double q1[20], qh[20], ql[20];
    size_t bw, bc;  // batched launch (crt_kernel, gridDim.z items): bytes between the items' workspaces / between their C matrices
};

// host side (oz2_crt.hip)
void fill_crt_tables(CrtArgs& a, int dtype, int backend, unsigned N);
void fill_crt_scalars(CrtArgs& a, int dtype, const void* alpha, const void* beta, bool scalars_on_device);

template <typename U> __device__ __forceinline__ U scalb(U x, int s);
template <> __device__ __forceinline__ float scalb<float>(float x, int s) { return scalbnf(x, s); }
template <> __device__ __forceinline__ double scalb<double>(double x, int s) { return scalbn(x, s); }
template <typename U> __device__ __forceinline__ U fmaU(U a, U b, U c);
template <> __device__ __forceinline__ float fmaU<float>(float a, float b, float c) { return fmaf(a, b, c); }
template <> __device__ __forceinline__ double fmaU<double>(double a, double b, double c) { return fma(a, b, c); }

__device__ __forceinline__ double crt_reduce(const CrtArgs& a, double Sh, double Sl) {
    const double q = rint(a.invP * Sh);
    if (!a.use_dd) return fma(a.Phi, q, Sh);
    return fma(a.Plo, q, fma(a.Phi, q, Sh) + Sl);
}

}  // namespace oz2
