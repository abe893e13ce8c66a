// FP6 (e2m3) operands of v_mfma_scale_f32_16x16x128_f8f6f4 (cbsz = blgp = 2), checked on the device (round 5):
//   1. field map: lane l = row / column l & 15, K group l >> 4; its 32 values are 6-bit fields, field f at bits 6 f .. 6 f + 5 of the
//      lane's six operand registers read as one 192-bit little-endian number
//   2. code map: sign << 5 | |v| is the e2m3 code of v / 8 for every integer |v| <= 16 (subnormal 0..7, normal 8..15, 16 = 2.0)
//   3. E8M0 scale 2^3 on both operands (0x82 in every byte) gives the integer product sums themselves
//   4. integer sums stay exact up to 2^24 in the accumulator (what the residue GEMMs of the FP8 backend need)
// build: hipcc --offload-arch=gfx950 -O2 f6_layout.hip -o f6_layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <cstdlib>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

// A, B: [16 rows][4 groups][6 dwords]; C0: accumulator start value (all elements); scale: E8M0 bytes
__global__ void k6(const unsigned* A, const unsigned* B, float* D, float c0, int scale) {
    const int l = threadIdx.x, q = l >> 4;
    auto frag = [&](const unsigned* M) {
        const unsigned* p = M + ((l & 15) * 4 + q) * 6;
        return v8i{(int)p[0], (int)p[1], (int)p[2], (int)p[3], (int)p[4], (int)p[5], 0, 0};
    };
    v4f c = {c0, c0, c0, c0};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(frag(A), frag(B), c, 2, 2, 0, scale, 0, scale);
    for (int r = 0; r < 4; ++r) D[(4 * q + r) * 16 + (l & 15)] = c[r];
}

static unsigned code(int v) { return (v < 0 ? 32u : 0u) | (unsigned)abs(v); }
static void put(unsigned* w, int f, unsigned c) {
    const int bit = 6 * f;
    w[bit >> 5] |= c << (bit & 31);
    if ((bit & 31) > 26) w[(bit >> 5) + 1] |= c >> (32 - (bit & 31));
}

int main() {
    unsigned *dA, *dB;
    float* dD;
    hipMalloc(&dA, 16 * 4 * 6 * 4), hipMalloc(&dB, 16 * 4 * 6 * 4), hipMalloc(&dD, 256 * 4);
    unsigned hA[16][4][6], hB[16][4][6];
    float hD[256];
    int bad = 0;
    auto run = [&](float c0, int scale) {
        hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice), hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k6, dim3(1), dim3(64), 0, 0, dA, dB, dD, c0, scale);
        hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
    };
    // 1 + 2: one non-zero field of A at (row r, group g, field f) = code(v); B = all fields code(8) (= 1.0): D[r][*] = v / 8
    for (int g = 0; g < 4; ++g)
        for (int f = 0; f < 32; ++f) {
            memset(hA, 0, sizeof hA), memset(hB, 0, sizeof hB);
            int vals[16];
            for (int r = 0; r < 16; ++r) {
                vals[r] = ((r * 7 + f * 3 + g) % 33) - 16;
                put(hA[r][g], f, code(vals[r]));
            }
            for (int c = 0; c < 16; ++c)
                for (int gg = 0; gg < 4; ++gg)
                    for (int ff = 0; ff < 32; ++ff) put(hB[c][gg], ff, code(8));
            run(0.f, 0x7F7F7F7F);
            for (int r = 0; r < 16; ++r)
                for (int c = 0; c < 16; ++c)
                    if (hD[r * 16 + c] != vals[r] / 8.0f) {
                        if (bad < 10) printf("field map: g %d f %d r %d c %d: got %g want %g\n", g, f, r, c, hD[r * 16 + c], vals[r] / 8.0f);
                        ++bad;
                    }
        }
    printf("field / code map: %s\n", bad ? "MISMATCH" : "ok (field f of group g at bits 6f..6f+5; code = sign << 5 | |v|, value v / 8)");
    // 3: random integers in [-16, 16], scales 2^3 x 2^3
    int bad3 = 0;
    unsigned s = 777;
    for (int rep = 0; rep < 50; ++rep) {
        int a[16][128], b[16][128];
        memset(hA, 0, sizeof hA), memset(hB, 0, sizeof hB);
        for (int r = 0; r < 16; ++r)
            for (int k = 0; k < 128; ++k) {
                s = s * 1664525u + 1013904223u, a[r][k] = (int)((s >> 8) % 33u) - 16;
                s = s * 1664525u + 1013904223u, b[r][k] = (int)((s >> 8) % 33u) - 16;
                put(hA[r][k >> 5], k & 31, code(a[r][k]));
                put(hB[r][k >> 5], k & 31, code(b[r][k]));
            }
        run(0.f, (int)0x82828282u);
        for (int r = 0; r < 16; ++r)
            for (int c = 0; c < 16; ++c) {
                int e = 0;
                for (int k = 0; k < 128; ++k) e += a[r][k] * b[c][k];
                if (hD[r * 16 + c] != (float)e) {
                    if (bad3 < 10) printf("random: rep %d r %d c %d: got %g want %d\n", rep, r, c, hD[r * 16 + c], e);
                    ++bad3;
                }
            }
    }
    printf("random integers in [-16, 16], scales 2^3: %s\n", bad3 ? "MISMATCH" : "ok (exact)");
    // 4: accumulator exactness: C_in = +-(2^p - 1 - 128 * 256) .. with block sums of extreme size
    int bad4 = 0;
    for (int p = 18; p <= 24; ++p)
        for (int sign = -1; sign <= 1; sign += 2)
            for (int mag : {1, 16}) {
                memset(hA, 0, sizeof hA), memset(hB, 0, sizeof hB);
                // block sum = 128 * mag * mag' with one field = mag and the partner 1 (sum = mag), or all fields (sum = 128 mag^2)
                for (int r = 0; r < 16; ++r) put(hA[r][3], 31, code(mag)), put(hB[r][3], 31, code(1));
                const float c0 = sign * (float)((1 << p) - 1 - mag);
                run(c0, (int)0x82828282u);
                const double want = (double)c0 + mag;
                for (int i = 0; i < 256; ++i)
                    if ((double)hD[i] != want) {
                        if (bad4 < 10) printf("accumulate: p %d sign %d mag %d: got %.1f want %.1f\n", p, sign, mag, hD[i], want);
                        ++bad4;
                        break;
                    }
                for (int r = 0; r < 16; ++r)
                    for (int g = 0; g < 4; ++g)
                        for (int f = 0; f < 32; ++f) {
                            if (g == 3 && f == 31) continue;
                            put(hA[r][g], f, code(mag)), put(hB[r][g], f, code(-mag));
                        }
                // 127 products -mag^2 and one +mag
                const float c1 = sign * (float)((1 << p) - 1 - 127 * mag * mag);
                run(c1, (int)0x82828282u);
                const double want1 = (double)c1 + mag - 127.0 * mag * mag;
                for (int i = 0; i < 256; ++i)
                    if ((double)hD[i] != want1) {
                        if (bad4 < 10) printf("accumulate(full): p %d sign %d mag %d: got %.1f want %.1f\n", p, sign, mag, hD[i], want1);
                        ++bad4;
                        break;
                    }
            }
    printf("integer accumulation up to 2^24: %s\n", bad4 ? "MISMATCH" : "ok (exact)");
    return (bad || bad3 || bad4) ? 1 : 0;
}
