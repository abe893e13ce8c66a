#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef unsigned v6u __attribute__((ext_vector_type(6)));
__global__ void k2(const float* in, unsigned* out, float scale) {
    v16f a, b;
    for (int i = 0; i < 16; ++i) a[i] = in[threadIdx.x * 32 + i], b[i] = in[threadIdx.x * 32 + 16 + i];
    v6u r = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(a, b, scale);
    for (int i = 0; i < 6; ++i) out[threadIdx.x * 6 + i] = r[i];
}
int main() {
    float h[64 * 32];
    for (int l = 0; l < 64; ++l)
        for (int i = 0; i < 32; ++i) h[l * 32 + i] = l == 0 ? (float)(i - 16) : l == 1 ? (float)(i) * 0.5f : l == 2 ? (i < 16 ? 100.f + i : -(100.f + i)) : (float)((i * 7 + l) % 33 - 16);
    float* d; unsigned* o; unsigned ho[64 * 6];
    hipMalloc(&d, sizeof h); hipMalloc(&o, sizeof ho);
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    const float scales[3] = {1.0f, 8.0f, 0.125f};
    for (int s = 0; s < 3; ++s) {
        k2<<<1, 64>>>(d, o, scales[s]);
        hipMemcpy(ho, o, sizeof ho, hipMemcpyDeviceToHost);
        for (int l = 0; l < 4; ++l) {
            printf("scale %g lane %d codes:", scales[s], l);
            for (int i = 0; i < 32; ++i) {
                unsigned bit = i * 6, w = bit / 32, sh = bit % 32;
                unsigned long long two = ho[l * 6 + w] | ((w + 1 < 6 ? (unsigned long long)ho[l * 6 + w + 1] : 0ull) << 32);
                printf(" %02x", (unsigned)((two >> sh) & 63));
            }
            printf("\n");
        }
    }
    return 0;
}
