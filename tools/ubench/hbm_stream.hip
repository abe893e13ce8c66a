// Achievable HBM streaming rates of hand-written 16-byte-per-lane kernels on one MI355X: the yardstick for the HBM-bound kernels of the
// emulation (quantise, CRT, amax / extract).  MI355X_MICROARCH.md quotes 6.29 TB/s for a float4 copy; torch.copy_ reaches 4.96.
//   copy / read / write over 2 GiB, plain and non-temporal, 1..4 independent 16-byte accesses in flight per lane and loop trip;
//   "crtmix": the CRT kernel's traffic shape without its arithmetic -- 14 planes of 64 MiB read, 512 MiB written.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned V4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int U, bool NT> __global__ void __launch_bounds__(256) copy_k(const V4* __restrict__ src, V4* __restrict__ dst, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i + (U - 1) * stride < n; i += U * stride) {
        V4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (NT) __builtin_nontemporal_store(v[u], dst + i + u * stride);
            else dst[i + u * stride] = v[u];
        }
    }
}
template <int U, bool NT> __global__ void __launch_bounds__(256) read_k(const V4* __restrict__ src, unsigned* out, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256;
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i + (U - 1) * stride < n; i += U * stride) {
        V4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
template <bool NT> __global__ void __launch_bounds__(256) write_k(V4* __restrict__ dst, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const V4 v = {(unsigned)i, 1u, 2u, 3u};
        if (NT) __builtin_nontemporal_store(v, dst + i);
        else dst[i] = v;
    }
}
// one thread: 8 bytes of each of 14 planes (plane stride = elements), 64 bytes out
template <int MODE> __global__ void __launch_bounds__(256) crtmix_k(const unsigned long long* __restrict__ planes, size_t plane_stride, V4* __restrict__ dst, size_t nthreads) {
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= nthreads) return;
    unsigned long long c[14];
#pragma unroll
    for (int t = 0; t < 14; ++t) c[t] = __builtin_nontemporal_load(planes + t * plane_stride + gid);
    unsigned long long x = 0;
#pragma unroll
    for (int t = 0; t < 14; ++t) x += c[t] * (t + 1);
    V4 o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = V4{(unsigned)x + j, (unsigned)(x >> 32), (unsigned)j, 7u};
    if (MODE == 0) {  // lane-strided: thread writes its own 64 contiguous bytes
#pragma unroll
        for (int j = 0; j < 4; ++j) __builtin_nontemporal_store(o[j], dst + gid * 4 + j);
    } else {          // lane-linear: instruction j writes 1 KiB of contiguous memory per wave (values permuted: traffic only)
        const size_t wbase = (gid & ~(size_t)63) * 4;
        const unsigned lane = threadIdx.x & 63;
#pragma unroll
        for (int j = 0; j < 4; ++j) __builtin_nontemporal_store(o[j], dst + wbase + j * 64 + lane);
    }
}

// the quantise kernel's traffic shape: 32 bytes in per lane (4 doubles), one dword out to each of NP planes `plane_stride` bytes apart
template <int NP> __global__ void __launch_bounds__(256) quantmix_k(const V4* __restrict__ src, char* __restrict__ dst, size_t plane_stride, size_t nthreads) {
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= nthreads) return;
    const V4 a = __builtin_nontemporal_load(src + 2 * gid), b = __builtin_nontemporal_load(src + 2 * gid + 1);
    unsigned x = a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w;
#pragma unroll
    for (int t = 0; t < NP; ++t) {
        x = x * 1664525u + 1013904223u;
        *(unsigned*)(dst + (size_t)t * plane_stride + gid * 4) = x;
    }
}
// the CRT shape with a parameterised plane stride (bytes): 8 bytes of each of 14 planes in, 64 bytes out lane-linear
__global__ void __launch_bounds__(256) crtmix_stride_k(const char* __restrict__ planes, size_t plane_stride, V4* __restrict__ dst, size_t nthreads) {
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= nthreads) return;
    unsigned long long c[14];
#pragma unroll
    for (int t = 0; t < 14; ++t) c[t] = __builtin_nontemporal_load((const unsigned long long*)(planes + t * plane_stride) + gid);
    unsigned long long x = 0;
#pragma unroll
    for (int t = 0; t < 14; ++t) x += c[t] * (t + 1);
    const size_t wbase = (gid & ~(size_t)63) * 4;
    const unsigned lane = threadIdx.x & 63;
#pragma unroll
    for (int j = 0; j < 4; ++j) __builtin_nontemporal_store(V4{(unsigned)x + j, (unsigned)(x >> 32), (unsigned)j, 7u}, dst + wbase + j * 64 + lane);
}

int main() {
    const size_t bytes = (size_t)2 << 30, n = bytes / 16;
    V4 *a, *b;
    unsigned* flag;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&flag, 4));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](auto launch, double moved, const char* name) {
        for (int i = 0; i < 3; ++i) launch();
        hipEventRecord(e0);
        const int reps = 10;
        for (int i = 0; i < reps; ++i) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-44s %8.1f us  %6.2f TB/s\n", name, ms / reps * 1e3, moved / (ms / reps * 1e-3) * 1e-12);
    };
    for (int blocks : {2048, 8192, 32768}) {
        printf("-- grid %d x 256\n", blocks);
        time([&] { copy_k<1, false><<<blocks, 256>>>(a, b, n); }, 2.0 * bytes, "copy 16 B/lane x1");
        time([&] { copy_k<4, false><<<blocks, 256>>>(a, b, n); }, 2.0 * bytes, "copy 16 B/lane x4");
        time([&] { copy_k<4, true><<<blocks, 256>>>(a, b, n); }, 2.0 * bytes, "copy 16 B/lane x4 nt");
        time([&] { read_k<4, false><<<blocks, 256>>>(a, flag, n); }, 1.0 * bytes, "read 16 B/lane x4");
        time([&] { read_k<4, true><<<blocks, 256>>>(a, flag, n); }, 1.0 * bytes, "read 16 B/lane x4 nt");
        time([&] { write_k<false><<<blocks, 256>>>(b, n); }, 1.0 * bytes, "write 16 B/lane");
        time([&] { write_k<true><<<blocks, 256>>>(b, n); }, 1.0 * bytes, "write 16 B/lane nt");
    }
    // CRT traffic shape at config 2: 8192 x 8192 outputs, 14 planes of 64 MiB, 512 MiB out
    const size_t nthr = (size_t)8192 * 8192 / 8, pstride = (size_t)8192 * 8192 / 8;  // in 8-byte units
    if (14 * pstride * 8 <= bytes) {
        const double moved = 14.0 * pstride * 8 + nthr * 64.0;
        time([&] { crtmix_k<0><<<(unsigned)(nthr / 256), 256>>>((const unsigned long long*)a, pstride, b, nthr); }, moved, "crtmix 14 x 8 B in, 64 B out lane-strided");
        time([&] { crtmix_k<1><<<(unsigned)(nthr / 256), 256>>>((const unsigned long long*)a, pstride, b, nthr); }, moved, "crtmix 14 x 8 B in, 64 B out lane-linear");
    }
    // plane-stride experiments: 8192 x 8192 elements, planes of 64 MiB; stride = 64 MiB + pad
    {
        const size_t nel = (size_t)8192 * 8192;
        char* big;
        const size_t plane = nel;  // bytes
        CK(hipMalloc(&big, 15 * (plane + (1 << 20))));
        CK(hipMemset(big, 3, 15 * (plane + (1 << 20))));
        for (size_t pad : {(size_t)0, (size_t)256, (size_t)4096, (size_t)8192, (size_t)(8192 + 256), (size_t)65536 + 4096 + 256, (size_t)(1 << 20) - 256}) {
            char name[128];
            snprintf(name, sizeof name, "quantmix 32 B in, 14 x 4 B out, stride 64 MiB + %zu", pad);
            time([&] { quantmix_k<14><<<(unsigned)(nel / 4 / 256), 256>>>(a, big, plane + pad, nel / 4); }, 8.0 * nel + 14.0 * nel, name);
            snprintf(name, sizeof name, "crtmix   14 x 8 B in, 64 B out, stride 64 MiB + %zu", pad);
            time([&] { crtmix_stride_k<<<(unsigned)(nel / 8 / 256), 256>>>(big, plane + pad, b, nel / 8); }, 8.0 * nel + 14.0 * nel, name);
        }
    }
    return 0;
}
