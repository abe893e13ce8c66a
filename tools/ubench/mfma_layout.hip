// Operand / result lane maps of v_mfma_i32_16x16x64_i8 and the semantics of v_permlane16_swap / v_permlane32_swap, checked on the
// device against a host product (the consumer loop and epilogues of oz2_gemm_i8.hip are built on exactly these assumptions):
//   A operand (v4i): lane l holds row i = l & 15, K bytes 16 * (l >> 4) .. + 15;  B operand: same with column j = l & 15
//   result (v4i)   : lane l holds column j = l & 15, rows i = 4 * (l >> 4) + r for register r = 0..3
//   permlane32_swap(x, y) -> {x', y'}: x' = lanes 0-31 keep x, lanes 32-63 get y of lane - 32; y' = lanes 0-31 get x of lane + 32, lanes 32-63 keep y
//   permlane16_swap(x, y) -> the same per pair of 16-lane rows: x' = even rows keep x, odd rows get y of lane - 16; y' = even rows get x of lane + 16, odd rows keep y
// build: hipcc --offload-arch=gfx950 -O2 mfma_layout.hip -o mfma_layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

// v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3 x e4m3, unit E8M0 scales): A / B operand (v8i) of lane l = row / column l & 15, 32 K values;
// result as for the INT8 form.  Any assignment of the 128 K positions to (lane >> 4, byte) works as long as A and B use the same
// one: the FP8 GEMM gives lane quad q the 16-byte chunks q and q + 4 of the 128-byte K-step (conflict-free LDS reads).
__global__ void kf8(const unsigned char* A, const unsigned char* B, float* D) {
    const int l = threadIdx.x, q = l >> 4;
    auto frag = [&](const unsigned char* M) {
        const v4i lo = *(const v4i*)(M + (l & 15) * 128 + 16 * q);
        const v4i hi = *(const v4i*)(M + (l & 15) * 128 + 16 * (q + 4));
        return v8i{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };
    v4f c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(frag(A), frag(B), c, 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
    for (int r = 0; r < 4; ++r) D[(4 * q + r) * 16 + (l & 15)] = c[r];
}

__global__ void k(const int8_t* A, const int8_t* B, int* D, unsigned* P) {
    const int l = threadIdx.x;
    v4i a = *(const v4i*)(A + (l & 15) * 64 + 16 * (l >> 4));
    v4i b = *(const v4i*)(B + (l & 15) * 64 + 16 * (l >> 4));
    v4i c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * (l >> 4) + r) * 16 + (l & 15)] = c[r];
    unsigned x = 1000 + l, y = 2000 + l;
    auto s32 = __builtin_amdgcn_permlane32_swap(x, y, false, false);
    auto s16 = __builtin_amdgcn_permlane16_swap(x, y, false, false);
    P[l] = s32[0], P[64 + l] = s32[1], P[128 + l] = s16[0], P[192 + l] = s16[1];
}

int main() {
    int8_t hA[16 * 64], hB[16 * 64];
    unsigned s = 12345;
    for (auto& v : hA) s = s * 1664525u + 1013904223u, v = (int8_t)(s >> 24);
    for (auto& v : hB) s = s * 1664525u + 1013904223u, v = (int8_t)(s >> 24);
    int8_t *dA, *dB;
    int* dD;
    unsigned* dP;
    hipMalloc(&dA, sizeof hA), hipMalloc(&dB, sizeof hB), hipMalloc(&dD, 256 * 4), hipMalloc(&dP, 256 * 4);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice), hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD, dP);
    int hD[256];
    unsigned hP[256];
    hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost), hipMemcpy(hP, dP, sizeof hP, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            int ref = 0;
            for (int kk = 0; kk < 64; ++kk) ref += (int)hA[i * 64 + kk] * hB[j * 64 + kk];
            bad += ref != hD[i * 16 + j];
        }
    printf("mfma_i32_16x16x64_i8 lane maps: %s (%d of 256 differ)\n", bad ? "WRONG" : "as assumed", bad);
    int b32 = 0, b16 = 0;
    for (int l = 0; l < 64; ++l) {
        const unsigned e0 = l < 32 ? 1000 + l : 2000 + (l - 32), e1 = l < 32 ? 1000 + (l + 32) : 2000 + l;
        b32 += hP[l] != e0 || hP[64 + l] != e1;
        const bool odd = (l >> 4) & 1;
        const unsigned f0 = !odd ? 1000 + l : 2000 + (l - 16), f1 = !odd ? 1000 + (l + 16) : 2000 + l;
        b16 += hP[128 + l] != f0 || hP[192 + l] != f1;
    }
    printf("permlane32_swap: %s, permlane16_swap: %s\n", b32 ? "WRONG" : "as assumed", b16 ? "WRONG" : "as assumed");
    if (b16) {
        printf("permlane16_swap x':");
        for (int l = 0; l < 64; ++l) printf(" %u", hP[128 + l]);
        printf("\npermlane16_swap y':");
        for (int l = 0; l < 64; ++l) printf(" %u", hP[192 + l]);
        printf("\n");
    }
    // FP8: integers in [-16, 16] as e4m3 bytes (sign, 4-bit exponent bias 7, 3-bit mantissa)
    auto e4m3 = [](int v) -> unsigned char {
        if (v == 0) return 0;
        const unsigned sgn = v < 0 ? 0x80u : 0u;
        int a = v < 0 ? -v : v, e = 0;
        while ((a >> (e + 1)) != 0) ++e;               // a in [2^e, 2^(e+1))
        const int mant = ((a << 3) >> e) & 7;           // exact for a <= 16
        return (unsigned char)(sgn | ((unsigned)(e + 7) << 3) | (unsigned)mant);
    };
    unsigned char fA[16 * 128], fB[16 * 128];
    int iA[16 * 128], iB[16 * 128];
    for (int i = 0; i < 16 * 128; ++i) {
        s = s * 1664525u + 1013904223u, iA[i] = (int)((s >> 16) % 33u) - 16, fA[i] = e4m3(iA[i]);
        s = s * 1664525u + 1013904223u, iB[i] = (int)((s >> 16) % 33u) - 16, fB[i] = e4m3(iB[i]);
    }
    unsigned char *dfA, *dfB;
    float* dF;
    hipMalloc(&dfA, sizeof fA), hipMalloc(&dfB, sizeof fB), hipMalloc(&dF, 256 * 4);
    hipMemcpy(dfA, fA, sizeof fA, hipMemcpyHostToDevice), hipMemcpy(dfB, fB, sizeof fB, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(kf8, dim3(1), dim3(64), 0, 0, dfA, dfB, dF);
    float hF[256];
    hipMemcpy(hF, dF, sizeof hF, hipMemcpyDeviceToHost);
    int badf = 0;
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            int ref = 0;
            for (int kk = 0; kk < 128; ++kk) ref += iA[i * 128 + kk] * iB[j * 128 + kk];
            badf += (float)ref != hF[i * 16 + j];
        }
    printf("mfma_scale_f32_16x16x128_f8f6f4 (e4m3, unit scales) lane maps: %s (%d of 256 differ)\n", badf ? "WRONG" : "as assumed", badf);
    return bad || b32 || b16 || badf;
}
