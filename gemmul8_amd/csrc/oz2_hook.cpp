// LD_PRELOAD hipBLAS hook: hipblas{S,D,C,Z}gemm, hipblasGemmEx, their strided-batched forms, hipblasLtMatmul and hipblasDestroy are intercepted
// and routed to the Ozaki-II emulation (C ABI, gemmul8_c.h) according to the GEMMUL8_* environment
// variables; everything else -- and every call the environment does not select -- is passed to the
// real library found with dlsym(RTLD_NEXT).
//
// Behavioural contract restated from the reference (GEMMul8/src/hook.cu, README.md:283-385):
//   env (read on EVERY call unless noted)     hook.cu:170-227,284-310
//     GEMMUL8_BACKEND          0|INT8 (default) / 1|FP8
//     GEMMUL8_NUM_MOD_{D,S,Z,C} emulate when 2 <= N <= 20 (D,Z) / 13 (S,C); otherwise native
//     GEMMUL8_FASTMODE_{D,S,Z,C} "1" = fast mode, default accurate
//     GEMMUL8_SKIP_SCALE_{A,B}  "1" = keep quantised operand + shifts between calls (pointer identity)
//     GEMMUL8_MAX_{M,N,K}, GEMMUL8_MAX_NUM_MOD, GEMMUL8_MAXWS_BACKEND (read once): workspace
//       pre-sizing applied when a SKIP_SCALE switch is on                      hook.cu:232-281,656-662
//   early outs: m|n|k <= 0 -> SUCCESS, null A/B/C -> INVALID_VALUE            hook.cu:616-617
//   GEMMUL8_DIST (this build only)  blocks | moduli | fp64sum: shard every emulated GEMM over the ranks of an SPMD job (see try_dist)
//   GEMMUL8_MIN_FLOPS (this build only) UNSET or 0 = emulate every selected call (the reference's behaviour, hook.cu:600-660);
//                     "auto" = a fitted cost model decides per call whether the emulation wins (below_floor); a number = calls with
//                     2*m*n*k below it use the native routine.  The first call a floor hands to the native routine is logged once.
//   per-handle state under a mutex: three grow-only stream-ordered buffers (hipMallocAsync /
//   hipFreeAsync), event hand-off when the handle's stream changes, skip-scaling cache
//   (hook.cu:70-162,331-374,684-727); hipblasDestroy frees the state first (hook.cu:846-856).
//   GEMMUL8_HOOK_ROCBLAS=1 (this build only) also intercept rocblas_{s,d,c,z}gemm, their strided-batched forms and rocblas_gemm_ex: callers
//                     that use rocBLAS directly (HPL-style codes) -- NOT rocSOLVER, whose factorizations call rocBLAS's internal C++ templates
//                     (INTEGRATION.md "What the hook reaches")
//   GEMMUL8_HOOK_STATS=1 print at exit how many GEMM calls / flops were emulated and how many went to the native routines
// This file has no kernels and calls no BLAS routine itself.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <hipblas/hipblas.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <unordered_map>
#include <utility>
#include <vector>

#include "../../include/gemmul8_c.h"
#include "../../include/gemmul8_dist.h"

#ifdef OZ2_HOOK_SHIM
// Preload shim for hosts that load their HIP runtime late and privately (Python/PyTorch): this library contains no device
// code (so nothing registers with the HIP runtime when the loader maps it at process start) and binds to libgemmul8.so --
// which does -- on the first intercepted call: by then the host's libamdhip64 is mapped; it is promoted to the global symbol
// scope, then libgemmul8.so (found next to this file) is opened and the two C-ABI entry points the hook needs are resolved.
#include <dlfcn.h>
#include <string>
namespace {
struct Abi {
    size_t (*work_size)(int, int, size_t, size_t, size_t, unsigned, int, int, size_t*, size_t*) = nullptr;
    int (*gemm)(void*, int, int, int, int, size_t, size_t, size_t, const void*, const void*, size_t, const void*, size_t, const void*, void*,
                size_t, unsigned, int, void*, void*, void*, int, int, int, int, double*) = nullptr;
    decltype(&::gemmul8_work_size_batched) work_size_batched = nullptr;
    decltype(&::gemmul8_gemm_batched) gemm_batched = nullptr;
    decltype(&::gemmul8_add_row_bias) add_row_bias = nullptr;
    decltype(&::gemmul8_comm_rccl_from_env) comm_from_env = nullptr;
    decltype(&::gemmul8_dist_create) dist_create = nullptr;
    decltype(&::gemmul8_dist_gemm) dist_gemm = nullptr;
    decltype(&::gemmul8_dist_allgather_c) dist_allgather_c = nullptr;
    decltype(&::gemmul8_dist_destroy) dist_destroy = nullptr;
    decltype(&::gemmul8_set_fp8_bound_mode) set_fp8_bound_mode = nullptr;
};
const Abi& abi() {
    static const Abi a = [] {
        Abi r;
        if (FILE* f = std::fopen("/proc/self/maps", "r")) {  // promote the HIP runtime the process already uses
            char line[1024];
            while (std::fgets(line, sizeof line, f)) {
                if (!std::strstr(line, "libamdhip64")) continue;
                char* path = std::strchr(line, '/');
                if (!path) continue;
                path[std::strcspn(path, "\n")] = 0;
                (void)dlopen(path, RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);
                break;
            }
            std::fclose(f);
        }
        Dl_info info;
        std::string dir = ".";
        if (dladdr((const void*)&abi, &info) && info.dli_fname) {
            dir = info.dli_fname;
            const size_t slash = dir.rfind('/');
            dir = slash == std::string::npos ? "." : dir.substr(0, slash);
        }
        void* h = dlopen((dir + "/libgemmul8.so").c_str(), RTLD_NOW | RTLD_LOCAL);
        if (!h) {
            std::fprintf(stderr, "[GEMMUL8 HOOK] cannot open %s/libgemmul8.so: %s\n", dir.c_str(), dlerror());
            std::abort();
        }
        r.work_size = (decltype(r.work_size))dlsym(h, "gemmul8_work_size");
        r.gemm = (decltype(r.gemm))dlsym(h, "gemmul8_gemm");
        r.work_size_batched = (decltype(r.work_size_batched))dlsym(h, "gemmul8_work_size_batched");
        r.gemm_batched = (decltype(r.gemm_batched))dlsym(h, "gemmul8_gemm_batched");
        r.add_row_bias = (decltype(r.add_row_bias))dlsym(h, "gemmul8_add_row_bias");
        r.comm_from_env = (decltype(r.comm_from_env))dlsym(h, "gemmul8_comm_rccl_from_env");
        r.dist_create = (decltype(r.dist_create))dlsym(h, "gemmul8_dist_create");
        r.dist_gemm = (decltype(r.dist_gemm))dlsym(h, "gemmul8_dist_gemm");
        r.dist_allgather_c = (decltype(r.dist_allgather_c))dlsym(h, "gemmul8_dist_allgather_c");
        r.dist_destroy = (decltype(r.dist_destroy))dlsym(h, "gemmul8_dist_destroy");
        r.set_fp8_bound_mode = (decltype(r.set_fp8_bound_mode))dlsym(h, "gemmul8_set_fp8_bound_mode");
        if (!r.work_size || !r.gemm || !r.work_size_batched || !r.gemm_batched || !r.add_row_bias || !r.comm_from_env || !r.dist_create || !r.dist_gemm || !r.dist_allgather_c || !r.dist_destroy || !r.set_fp8_bound_mode) {
            std::fprintf(stderr, "[GEMMUL8 HOOK] libgemmul8.so lacks the C ABI entry points\n");
            std::abort();
        }
        return r;
    }();
    return a;
}
}  // namespace
#define gemmul8_work_size abi().work_size
#define gemmul8_gemm abi().gemm
#define gemmul8_work_size_batched abi().work_size_batched
#define gemmul8_gemm_batched abi().gemm_batched
#define gemmul8_add_row_bias abi().add_row_bias
#define gemmul8_comm_rccl_from_env abi().comm_from_env
#define gemmul8_dist_create abi().dist_create
#define gemmul8_dist_gemm abi().dist_gemm
#define gemmul8_dist_allgather_c abi().dist_allgather_c
#define gemmul8_dist_destroy abi().dist_destroy
#define gemmul8_set_fp8_bound_mode abi().set_fp8_bound_mode
#endif

namespace {

struct Cache {  // what the quantised planes in workA/workB currently hold
    bool valid = false;
    unsigned num_moduli = 0;
    int op_A = 0, op_B = 0;
    size_t m = 0, n = 0, k = 0, lda = 0, ldb = 0;
    const void *A = nullptr, *B = nullptr;
    void *workA = nullptr, *workB = nullptr;
    int dtype = -1, backend = 0;
    bool fastmode = false;
    bool enA = false, enB = false;  // the skip switches move the workspace carving (extra bound plane): part of the key
};

struct Buffer {
    void* ptr = nullptr;
    size_t size = 0;
};

// side lane of a batched call: its own stream, join event and single work buffer (see emulate_batch)
struct BatchLane {
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;
    Buffer w;
};
constexpr int kMaxBatchLanes = 8;

struct HandleState {
    std::mutex mtx;
    Buffer wA, wB, wC;
    Cache last;
    hipStream_t last_stream = nullptr;
    hipEvent_t last_event = nullptr;
    bool have_stream = false;
    bool is_lt = false;  // the key is a hipblasLtHandle_t: it has no stream of its own and must never reach hipblasGetStream
    BatchLane lanes[kMaxBatchLanes];  // lane 0 unused (= the handle's stream and buffers)
    hipEvent_t fork = nullptr;
};

std::mutex g_map_mtx;
std::unordered_map<hipblasHandle_t, std::shared_ptr<HandleState>> g_map;

std::shared_ptr<HandleState> state_of(hipblasHandle_t h) {
    std::lock_guard<std::mutex> g(g_map_mtx);
    auto& p = g_map[h];
    if (!p) p = std::make_shared<HandleState>();
    return p;
}

bool env_one(const char* name) {
    const char* s = std::getenv(name);
    return s && std::strcmp(s, "1") == 0;
}
unsigned long long env_u64(const char* name, unsigned long long def) {
    const char* s = std::getenv(name);
    if (!s || !*s) return def;
    char* end = nullptr;
    const unsigned long long v = std::strtoull(s, &end, 10);
    return (end == s) ? def : v;
}
int env_backend(const char* name, int def, bool allow_both) {
    const char* s = std::getenv(name);
    if (!s) return def;
    if (!std::strcmp(s, "0") || !std::strcmp(s, "INT8")) return 0;
    if (!std::strcmp(s, "1") || !std::strcmp(s, "FP8")) return 1;
    if (allow_both && (!std::strcmp(s, "2") || !std::strcmp(s, "BOTH"))) return 2;
    return def;
}

struct TypeInfo {
    const char* nmod;
    const char* fast;
    unsigned max_moduli;
    bool cplx;
};
const TypeInfo kTypes[4] = {
    {"GEMMUL8_NUM_MOD_S", "GEMMUL8_FASTMODE_S", 13u, false},
    {"GEMMUL8_NUM_MOD_D", "GEMMUL8_FASTMODE_D", 20u, false},
    {"GEMMUL8_NUM_MOD_C", "GEMMUL8_FASTMODE_C", 13u, true},
    {"GEMMUL8_NUM_MOD_Z", "GEMMUL8_FASTMODE_Z", 20u, true},
};

// process-wide workspace floor, computed once (hook.cu:232-281)
size_t g_maxA = 0, g_maxB = 0, g_maxC = 0;
std::once_flag g_max_once;
void init_max_workspace() {
    std::call_once(g_max_once, [] {
        // GEMMUL8_FP8_BOUND=reference (this build only): the reference's (k+1)*2^-24 inflation of the FP8 bound GEMM instead of the
        // engine-safe default (include/gemmul8_c.h, gemmul8_set_fp8_bound_mode)
        if (const char* fb = std::getenv("GEMMUL8_FP8_BOUND"))
            if (!std::strcmp(fb, "reference")) (void)gemmul8_set_fp8_bound_mode(1);
        const size_t mm = env_u64("GEMMUL8_MAX_M", 0), mn = env_u64("GEMMUL8_MAX_N", 0), mk = env_u64("GEMMUL8_MAX_K", 0);
        const unsigned mmod = (unsigned)env_u64("GEMMUL8_MAX_NUM_MOD", 2);
        const bool cplx = env_u64("GEMMUL8_NUM_MOD_Z", 0) > 0 || env_u64("GEMMUL8_NUM_MOD_C", 0) > 0;
        const int which = env_backend("GEMMUL8_MAXWS_BACKEND", 0, true);
        for (int be = 0; be < 2; ++be) {
            if (!(which == be || which == 2)) continue;
            if (mmod < 2 || mmod > 20) continue;
            size_t wa = 0, wb = 0;
            const size_t w = gemmul8_work_size(cplx, be, mm, mn, mk, mmod, 1, 1, &wa, &wb);
            g_maxA = std::max(g_maxA, wa);
            g_maxB = std::max(g_maxB, wb);
            g_maxC = std::max(g_maxC, w > wa + wb ? w - wa - wb : 0);
        }
    });
}

hipblasStatus_t grow(Buffer& b, size_t need, hipStream_t stream, const char* tag) {
    if (need == 0 || (b.ptr && b.size >= need)) return HIPBLAS_STATUS_SUCCESS;
    if (b.ptr) {
        const hipError_t e = hipFreeAsync(b.ptr, stream);
        if (e != hipSuccess) {
            std::fprintf(stderr, "[GEMMUL8 HOOK] hipFreeAsync failed for %s (%s)\n", tag, hipGetErrorString(e));
            return HIPBLAS_STATUS_INTERNAL_ERROR;
        }
        b.ptr = nullptr;
        b.size = 0;
    }
    void* p = nullptr;
    const hipError_t e = hipMallocAsync(&p, need, stream);
    if (e != hipSuccess) {
        std::fprintf(stderr, "[GEMMUL8 HOOK] hipMallocAsync failed for %s size %zu bytes (%s)\n", tag, need, hipGetErrorString(e));
        return HIPBLAS_STATUS_ALLOC_FAILED;
    }
    b.ptr = p;
    b.size = need;
    return HIPBLAS_STATUS_SUCCESS;
}

// The real hipBLAS entry point: the next definition in the global search order, or -- when the host loaded hipBLAS privately
// (Python extension modules are dlopen'ed RTLD_LOCAL) -- the copy of libhipblas that is already mapped into the process.
void* mapped_library(const char* needle) {
    void* r = nullptr;
    if (FILE* f = std::fopen("/proc/self/maps", "r")) {
        char line[1024];
        while (std::fgets(line, sizeof line, f)) {
            if (!std::strstr(line, needle)) continue;
            char* path = std::strchr(line, '/');
            if (!path) continue;
            path[std::strcspn(path, "\n")] = 0;
            r = dlopen(path, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
            if (r) break;
        }
        std::fclose(f);
    }
    return r;
}
void* mapped_hipblas() {
    static void* h = mapped_library("libhipblas.so");  // not libhipblaslt
    return h;
}
void* mapped_hipblaslt() {
    static void* h = mapped_library("libhipblaslt.so");
    return h;
}
template <typename Fn> Fn real_fn(const char* name) {
    void* f = dlsym(RTLD_NEXT, name);
    if (!f)
        if (void* h = (std::strncmp(name, "hipblasLt", 9) == 0 ? mapped_hipblaslt() : mapped_hipblas())) f = dlsym(h, name);
    return reinterpret_cast<Fn>(f);
}

hipStream_t handle_stream(hipblasHandle_t h, hipblasStatus_t* st) {
    using Fn = hipblasStatus_t (*)(hipblasHandle_t, hipStream_t*);
    static Fn fn = [] {
        return real_fn<Fn>("hipblasGetStream");
    }();
    hipStream_t s = nullptr;
    *st = fn ? fn(h, &s) : HIPBLAS_STATUS_NOT_INITIALIZED;
    return s;
}

// order work on the new stream after everything queued on the previous one (hook.cu:141-162)
hipblasStatus_t order_streams(HandleState& st, hipStream_t cur) {
    if (!st.have_stream) {
        st.last_stream = cur;
        st.have_stream = true;
        return HIPBLAS_STATUS_SUCCESS;
    }
    if (st.last_stream == cur) return HIPBLAS_STATUS_SUCCESS;
    if (!st.last_event && hipEventCreateWithFlags(&st.last_event, hipEventDisableTiming) != hipSuccess) return HIPBLAS_STATUS_INTERNAL_ERROR;
    if (hipEventRecord(st.last_event, st.last_stream) != hipSuccess) return HIPBLAS_STATUS_INTERNAL_ERROR;
    if (hipStreamWaitEvent(cur, st.last_event, 0) != hipSuccess) return HIPBLAS_STATUS_INTERNAL_ERROR;
    st.last_stream = cur;
    return HIPBLAS_STATUS_SUCCESS;
}

// returns true and sets *status when the call was emulated; false -> caller passes through
// ---- GEMMUL8_DIST = blocks | moduli | fp64sum (not in the reference, which is single-GPU): an SPMD application -- every rank of a
// torchrun / mpirun job issuing the SAME GEMM calls on replicated operands, RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in its
// environment -- gets each emulated GEMM sharded over the ranks' GPUs by the plans of include/gemmul8_dist.h; the result is
// all-gathered so that every rank ends up with the full C, as it would without the hook.  One RCCL communicator per process, a
// small cache of plans keyed by the call's shape (a plan owns its workspaces).
struct DistKey {
    int kind, dtype, backend, ta, tb, fast;
    size_t m, n, k;
    unsigned N;
    bool operator==(const DistKey& o) const {
        return kind == o.kind && dtype == o.dtype && backend == o.backend && ta == o.ta && tb == o.tb && fast == o.fast && m == o.m && n == o.n &&
               k == o.k && N == o.N;
    }
};
// One communicator and one plan cache serve every handle and stream of the process, so the sharded calls are CHAINED: each call
// records `tail` on its stream when its last operation is enqueued and the next call -- on whatever stream -- first makes its
// stream wait for it.  Two handles / streams issuing same-shape GEMMs therefore never overlap on a plan's workspaces, and the RCCL
// operations of the one communicator execute in the order they were enqueued (mtx gives the enqueue order; every rank of an SPMD
// job issues the same sequence).
struct DistState {
    std::mutex mtx;
    gemmul8_comm* comm = nullptr;
    bool failed = false;
    std::vector<std::pair<DistKey, gemmul8_dist_plan*>> plans;  // most recently used last; at most 8
    hipEvent_t tail = nullptr;      // completion of the most recent sharded call
    hipStream_t tail_stream = nullptr;
    bool have_tail = false;
};
DistState g_dist;

int dist_kind_from_env() {
    const char* s = std::getenv("GEMMUL8_DIST");
    if (!s || !*s || !std::strcmp(s, "0")) return -1;
    if (!std::strcmp(s, "moduli")) return GEMMUL8_DIST_MODULI;
    if (!std::strcmp(s, "fp64sum")) return GEMMUL8_DIST_MODULI_FP64SUM;
    return GEMMUL8_DIST_BLOCKS;  // "1", "blocks"
}

// returns true when the sharded path took the call (status in *status); false -> single-GPU emulation.
// Falling back is only safe when EVERY rank takes the same decision, i.e. for reasons that are a function of the call's arguments
// and the environment (negative GEMMUL8_E_ARG / _NUM_MODULI / _UNSUPPORTED from plan creation, no communicator at all).  A
// resource failure on one rank (GEMMUL8_E_INTERNAL: allocation, transport) is reported as HIPBLAS_STATUS_INTERNAL_ERROR instead: a
// rank that silently computed alone would leave its peers waiting in a collective.
bool try_dist(int kind, int dtype, int backend, hipblasOperation_t ta, hipblasOperation_t tb, int m, int n, int k, const void* alpha, const void* A,
              int lda, const void* B, int ldb, const void* beta, void* C, int ldc, unsigned N, bool fastmode, hipStream_t stream,
              hipblasStatus_t* status) {
    std::lock_guard<std::mutex> lk(g_dist.mtx);
    if (g_dist.failed) return false;
    if (!g_dist.comm) {
        const int rc = gemmul8_comm_rccl_from_env(&g_dist.comm);
        if (rc != 0 || !g_dist.comm) {
            // decided before any collective has been issued; ncclCommInitRank itself fails on every rank when one is missing
            std::fprintf(stderr, "[GEMMUL8 HOOK] GEMMUL8_DIST is set but no RCCL communicator could be created (status %d; RANK/WORLD_SIZE/"
                                 "MASTER_ADDR/MASTER_PORT?): single-GPU emulation\n", rc);
            g_dist.failed = true;
            return false;
        }
    }
    if (!g_dist.tail && hipEventCreateWithFlags(&g_dist.tail, hipEventDisableTiming) != hipSuccess) return *status = HIPBLAS_STATUS_INTERNAL_ERROR, true;
    // chain behind the previous sharded call (any handle, any stream)
    if (g_dist.have_tail && g_dist.tail_stream != stream && hipStreamWaitEvent(stream, g_dist.tail, 0) != hipSuccess)
        return *status = HIPBLAS_STATUS_INTERNAL_ERROR, true;
    const DistKey key{kind, dtype, backend, (int)ta, (int)tb, fastmode ? 1 : 0, (size_t)m, (size_t)n, (size_t)k, N};
    gemmul8_dist_plan* plan = nullptr;
    for (size_t i = 0; i < g_dist.plans.size(); ++i)
        if (g_dist.plans[i].first == key) {
            plan = g_dist.plans[i].second;
            std::rotate(g_dist.plans.begin() + i, g_dist.plans.begin() + i + 1, g_dist.plans.end());
            break;
        }
    if (!plan) {
        if (g_dist.plans.size() >= 8) {  // plans own workspaces of the problem's size: keep only a few
            // every earlier sharded call is an ancestor of `tail`: once it has completed no stream still uses the evicted plan
            if (g_dist.have_tail && hipEventSynchronize(g_dist.tail) != hipSuccess) (void)hipDeviceSynchronize();
            gemmul8_dist_destroy(g_dist.plans.front().second);
            g_dist.plans.erase(g_dist.plans.begin());
        }
        const int rc = gemmul8_dist_create(g_dist.comm, nullptr, kind, 0, dtype, backend, (int)ta, (int)tb, (size_t)m, (size_t)n, (size_t)k, N,
                                           fastmode ? 1 : 0, &plan);
        if (rc == GEMMUL8_E_INTERNAL || rc > 0 || (rc == 0 && !plan)) {
            std::fprintf(stderr, "[GEMMUL8 HOOK] GEMMUL8_DIST: creating the plan failed on this rank (status %d): returning an error (the other "
                                 "ranks are entering the collective)\n", rc);
            return *status = HIPBLAS_STATUS_INTERNAL_ERROR, true;
        }
        if (rc != 0) return false;  // a property of the arguments (outside the emulator's range): every rank declines alike
        g_dist.plans.emplace_back(key, plan);
    }
    int rc = gemmul8_dist_gemm(plan, stream, alpha, A, (size_t)lda, B, (size_t)ldb, beta, C, (size_t)ldc);
    if (rc == 0) rc = gemmul8_dist_allgather_c(plan, stream, C, (size_t)ldc);
    if (hipEventRecord(g_dist.tail, stream) == hipSuccess) g_dist.tail_stream = stream, g_dist.have_tail = true;
    else rc = rc ? rc : 1;
    *status = rc == 0 ? HIPBLAS_STATUS_SUCCESS : HIPBLAS_STATUS_INTERNAL_ERROR;
    return true;
}

// GEMMUL8_MIN_FLOPS (not in the reference).  A drop-in must not make an application slower by default: small products are ten
// latency-bound launches (1024^3: 21 vs 48 TFLOPS native), and what a hooked solver issues most -- trailing updates with large m = n
// and small k, panel products with one small dimension -- pays the per-output cost of the scheme (N bytes of residues written, read
// and recombined per element) without the k to amortise it (DGEMM 8192^2 x 256: 45 vs 65 TFLOPS native; x 512: 71 vs 69; x 1024:
// 107 vs 70).  With GEMMUL8_MIN_FLOPS=auto the hook therefore evaluates a fitted cost model per call and emulates only where the
// emulation is predicted to win (opt-in: the numerics of a call then depend on its shape, and the constants were fitted on one
// MI355X pool -- the default stays the reference's: every selected call is emulated):
//     emulated  t_e = c0 + (a1 + b1 N)(m + n) k + (a2 + b2 N) m n + b3 N m n k      per (type, accurate / fast), N = number of moduli
//     native    t_n = d0 + d2 m n + d3 m n k                                        per type (the library at its normal rate)
//     emulate   iff t_e <= 0.95 t_n        (a strided batch: one set of launches, so c0 / d0 once and the rest times the batch)
// Constants: tools/fit_floor.py on profiles/sweeps/r03_floor_scan_{s,d,c,z}.csv (tools/floor_scan.py: 60-68 shapes x 3 N x 2 modes
// per type against the native routine on the same box; median model error 5-7 %).  On the scanned shapes the rule emulates 31-53 of
// 180-204 cases per type and mode, lets 0-2 marginal losses through (worst 1.05x the native time, one 1.19x), and the summed time is
// within 0.3-10 % of always picking the faster of the two (always-native: +25-45 %); repeated on a second box with the final binaries
// (r03_floor_scan2_*.csv, not used for the fit) it stays within 0.4-12 % (tests/test_hook_floor.py).  The FP8 backend is priced as
// the INT8 model x a per-type factor: the median time ratio at EQUAL moduli count over the same shape classes on the round-5 kernels (FP6 operand
// planes, fused three-product tile loop; tools/fp8_factor_scan.py, profiles/sweeps/r05_fp8_factor_{s,d,c,z}.csv: 1.74 / 1.90 / 2.04 / 2.12,
// range 1.3-3.1; 2.2 for every type until round 4).
// GEMMUL8_MIN_FLOPS unset / 0 = the reference's behaviour (emulate every call); any other number is a plain floor on 2*m*n*k per call.
// GEMMUL8_HOOK_STATS=1: how much of an application's GEMM work the hook reaches (tests/test_gpu_hook_reach.py)
// A call handed to the native routine comes back through the interposed layers below it (hipblasDgemm -> rocblas_dgemm ->
// rocblas_internal_gemm_template with GEMMUL8_HOOK_ROCBLAS=1): it is the SAME call, already declined by the same rules.  Every
// pass-through runs inside a NativeScope; a hooked entry reached inside one goes straight to its real routine, uncounted.
thread_local int tl_native_depth = 0;
struct NativeScope {
    NativeScope() { ++tl_native_depth; }
    ~NativeScope() { --tl_native_depth; }
    NativeScope(const NativeScope&) = delete;
    NativeScope& operator=(const NativeScope&) = delete;
};
struct HookStats {
    std::atomic<unsigned long long> emu_calls{0}, nat_calls{0};
    std::atomic<unsigned long long> emu_mflops{0}, nat_mflops{0};  // 2 m n k batch / 1e6 (x 4 for complex), rounded down
    static void dump();
    HookStats() { std::atexit(&HookStats::dump); }
};
HookStats& hook_stats() {
    static HookStats* st = new HookStats;  // leaked on purpose: dumped from atexit
    return *st;
}
void HookStats::dump() {
    if (!env_one("GEMMUL8_HOOK_STATS")) return;
    HookStats& h = hook_stats();
    std::fprintf(stderr, "[GEMMUL8 HOOK] stats: emulated %llu GEMM calls (%.3f TFLOP), native %llu GEMM calls through the hooked entry points (%.3f TFLOP)\n",
                 h.emu_calls.load(), h.emu_mflops.load() * 1e-6, h.nat_calls.load(), h.nat_mflops.load() * 1e-6);
}
void count_call(bool emulated, int dtype, double m, double n, double k, double batch = 1.0) {
    static const bool on = env_one("GEMMUL8_HOOK_STATS");
    if (!on) return;
    HookStats& h = hook_stats();
    const unsigned long long mf = (unsigned long long)(2.0 * m * n * k * batch * (dtype >= 2 ? 4.0 : 1.0) * 1e-6);
    (emulated ? h.emu_calls : h.nat_calls).fetch_add(1, std::memory_order_relaxed);
    (emulated ? h.emu_mflops : h.nat_mflops).fetch_add(mf, std::memory_order_relaxed);
}

struct FloorModel {
    double e[6];  // ms: 1, (m+n)k, N(m+n)k, mn, N mn, N mnk
    double n[3];  // ms: 1, mn, mnk
};
// generated by tools/fit_floor.py from profiles/sweeps/r06_floor_scan_*.csv (the round-6 kernels: accurate mode two launches shorter, short-k epilogue +3-5 %; round 4's fit: r04b_floor_scan_*.csv)
static const FloorModel kFloor[4][2] = {  // [S, D, C, Z][accurate, fast]
    {{{0.05373, 2.353e-09, 3.999e-10, 1.821e-09, 3.555e-10, 6.155e-13}, {0.02155, 4.627e-10, 1.407e-11}}, {{0.04309, 1.051e-09, 4.238e-10, 1.091e-09, 3.574e-10, 5.79e-13}, {0.02155, 4.627e-10, 1.407e-11}}},
    {{{0.04743, 3.922e-09, 4.089e-10, 1.218e-09, 5.024e-10, 5.943e-13}, {0.01043, 8.077e-10, 2.819e-11}}, {{0.03986, 1.88e-09, 4.137e-10, 4.544e-10, 4.8e-10, 5.863e-13}, {0.01043, 8.077e-10, 2.819e-11}}},
    {{{0.06931, 6.912e-09, 1.196e-09, 3.604e-09, 1.806e-09, 1.761e-12}, {0.01406, 2.65e-10, 5.735e-11}}, {{0.05653, 2.963e-09, 1.234e-09, 2.005e-09, 1.787e-09, 1.72e-12}, {0.01406, 2.65e-10, 5.735e-11}}},
    {{{0.07734, 1.242e-08, 1.326e-09, 1.842e-09, 2.131e-09, 1.734e-12}, {0.01304, 2.591e-10, 1.098e-10}}, {{0.06313, 7.525e-09, 1.292e-09, 9.713e-10, 2.174e-09, 1.625e-12}, {0.01304, 2.591e-10, 1.098e-10}}},
};
static bool floor_model_declines(int dtype, double m, double n, double k, unsigned N, bool fast, int backend, double batch) {
    const FloorModel& fm = kFloor[dtype][fast ? 1 : 0];
    const double mk = (m + n) * k, mn = m * n, mnk = mn * k, Nd = (double)N;
    double te = fm.e[0] + batch * ((fm.e[1] + fm.e[2] * Nd) * mk + (fm.e[3] + fm.e[4] * Nd) * mn + fm.e[5] * Nd * mnk);
    static const double kFp8Factor[4] = {1.75, 1.9, 2.05, 2.1};  // [S, D, C, Z]
    if (backend == GEMMUL8_FP8) te *= kFp8Factor[dtype];
    const double tn = fm.n[0] + batch * (fm.n[1] * mn + fm.n[2] * mnk);
    return te > 0.95 * tn;
}
// quiet = a query (gemmul8_hook_would_emulate), not a call: no log line
bool below_floor(int dtype, double m, double n, double k, unsigned N, bool fast, int backend, double batch = 1.0, bool quiet = false) {
    const char* s = std::getenv("GEMMUL8_MIN_FLOPS");
    if (!s || !*s) return false;  // the reference's behaviour: every selected call is emulated
    bool declined;
    const bool automatic = (s[0] == 'a' || s[0] == 'A');
    if (automatic) {
        declined = floor_model_declines(dtype, m, n, k, N, fast, backend, batch);
    } else {
        const unsigned long long f = env_u64("GEMMUL8_MIN_FLOPS", 0);
        declined = f && 2.0 * m * n * k < (double)f;
    }
    if (declined && !quiet) {
        static std::once_flag told;
        std::call_once(told, [&] {
            std::fprintf(stderr, "[GEMMUL8 HOOK] GEMMUL8_MIN_FLOPS=%s: a %cGEMM %.0f x %.0f x %.0f (batch %.0f, %u moduli%s) stays on the native routine -- "
                                 "calls below the floor are NOT emulated (this message is printed once)\n",
                         s, "SDCZ"[dtype], m, n, k, batch, N, backend == GEMMUL8_FP8 ? ", FP8 backend: cost x 1.75-2.1" : "");
        });
    }
    return declined;
}

}  // namespace
extern "C" GEMMUL8_API int gemmul8_hook_would_emulate(int dtype, int backend, size_t m, size_t n, size_t k, unsigned num_moduli, int fastmode,
                                                      size_t batch) {
    if (dtype < 0 || dtype > 3 || (backend != GEMMUL8_INT8 && backend != GEMMUL8_FP8) || batch == 0 || num_moduli < 2 ||
        num_moduli > kTypes[dtype].max_moduli)
        return GEMMUL8_E_ARG;
    if (m == 0 || n == 0 || k == 0) return 0;
    return below_floor(dtype, (double)m, (double)n, (double)k, num_moduli, fastmode != 0, backend, (double)batch, true) ? 0 : 1;
}
namespace {
// explicit_stream: hipblasLtMatmul carries its stream as an argument (a hipblasLt handle has none)
bool try_emulate_impl(int dtype, hipblasHandle_t handle, hipblasOperation_t ta, hipblasOperation_t tb, int m, int n, int k, const void* alpha,
                      const void* A, int lda, const void* B, int ldb, const void* beta, void* C, int ldc, hipblasStatus_t* status,
                      const hipStream_t* explicit_stream) {
    const TypeInfo& ti = kTypes[dtype];
    const unsigned N = (unsigned)env_u64(ti.nmod, 0);
    if (N < 2u || N > ti.max_moduli) return false;
    const bool fastmode = env_one(ti.fast);
    const bool enA = env_one("GEMMUL8_SKIP_SCALE_A"), enB = env_one("GEMMUL8_SKIP_SCALE_B");
    const int backend = env_backend("GEMMUL8_BACKEND", 0, false);
    if (backend == GEMMUL8_FP8) {
        // the FP8 backend exists for parity with the reference; on this chip the INT8 backend dominates it: three GEMMs per modulus (on FP6 codes
        // of the backend's integer pieces, 1.5x the INT8 kernel's rate since round 5) against one INT8 GEMM (profiles/sweeps/r05_types_backends.csv:
        // SGEMM 8192^3 193 vs 302 TFLOPS, native 153; DGEMM 101 vs 160, native 71).  Say so once.
        static std::once_flag told;
        std::call_once(told, [] {
            std::fprintf(stderr, "[GEMMUL8 HOOK] GEMMUL8_BACKEND=FP8: on MI355X the INT8 backend (GEMMUL8_BACKEND=0) is 1.6-2x faster at equal or better "
                                 "accuracy for S/D/C/Z; continuing with FP8 as requested\n");
        });
    }
    if (below_floor(dtype, m, n, k, N, fastmode, backend)) return false;

    auto sp = state_of(handle);
    std::lock_guard<std::mutex> lk(sp->mtx);
    if (explicit_stream) sp->is_lt = true;
    init_max_workspace();
    hipblasStatus_t st = HIPBLAS_STATUS_SUCCESS;
    hipStream_t stream = explicit_stream ? *explicit_stream : handle_stream(handle, &st);
    if (st != HIPBLAS_STATUS_SUCCESS) return *status = st, true;
    if ((st = order_streams(*sp, stream)) != HIPBLAS_STATUS_SUCCESS) return *status = st, true;

    const int dist_kind = dist_kind_from_env();
    if (dist_kind >= 0 && try_dist(dist_kind, dtype, backend, ta, tb, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, N, fastmode, stream, status))
        return true;

    size_t needA = 0, needB = 0;
    const size_t tot = gemmul8_work_size(ti.cplx, backend, (size_t)m, (size_t)n, (size_t)k, N, enA, enB, &needA, &needB);
    if (tot < needA + needB) return *status = HIPBLAS_STATUS_INVALID_VALUE, true;
    size_t reqA = needA, reqB = needB, reqC = tot - needA - needB;
    if (enA || enB) {  // keep buffers (hence cached planes) stable across differently sized calls
        if (enA) reqA = std::max(reqA, g_maxA);
        if (enB) reqB = std::max(reqB, g_maxB);
        reqC = std::max(reqC, g_maxC);
    }
    if ((st = grow(sp->wA, reqA, stream, "workA")) != HIPBLAS_STATUS_SUCCESS) return *status = st, true;
    if ((st = grow(sp->wB, reqB, stream, "workB")) != HIPBLAS_STATUS_SUCCESS) return *status = st, true;
    if ((st = grow(sp->wC, reqC, stream, "workC")) != HIPBLAS_STATUS_SUCCESS) return *status = st, true;

    const Cache& c = sp->last;
    bool skipA = false, skipB = false;
    if (c.valid && c.num_moduli == N && c.k == (size_t)k && c.dtype == dtype && c.fastmode == fastmode && c.backend == backend &&
        c.enA == enA && c.enB == enB) {
        skipA = enA && c.workA == sp->wA.ptr && c.A == A && c.m == (size_t)m && c.lda == (size_t)lda && c.op_A == (int)ta;
        skipB = enB && c.workB == sp->wB.ptr && c.B == B && c.n == (size_t)n && c.ldb == (size_t)ldb && c.op_B == (int)tb;
    }
    sp->last.valid = false;  // the call below overwrites the planes; the cache is re-validated only if it succeeds
    const int rc = gemmul8_gemm(stream, dtype, backend, (int)ta, (int)tb, (size_t)m, (size_t)n, (size_t)k, alpha, A, (size_t)lda, B,
                                (size_t)ldb, beta, C, (size_t)ldc, N, fastmode, sp->wC.ptr, sp->wA.ptr, sp->wB.ptr, enA, enB, skipA, skipB,
                                nullptr);
    if (rc < 0) {
        // a GEMMUL8_E_* status means "this call is outside what the emulator accepts" (k > 2^17, FP8 with k > 65536, a
        // combination that is not built, ...): nothing has been written to C yet, so the application's call is still valid
        // for the native routine -- pass it through instead of failing a call that works without the hook
        static std::once_flag warned;
        std::call_once(warned, [&] {
            std::fprintf(stderr, "[GEMMUL8 HOOK] emulation declined a call (status %d; type %d, backend %d, m=%d n=%d k=%d): using the native routine for such calls\n",
                         rc, dtype, backend, m, n, k);
        });
        return false;
    }
    if (rc != 0) return *status = HIPBLAS_STATUS_INTERNAL_ERROR, true;  // positive: a hipError_t from the runtime
    Cache& u = sp->last;
    u.valid = true;
    u.enA = enA, u.enB = enB;
    u.num_moduli = N;
    u.op_A = (int)ta, u.op_B = (int)tb;
    u.m = m, u.n = n, u.k = k, u.lda = lda, u.ldb = ldb;
    u.A = A, u.B = B;
    u.workA = sp->wA.ptr, u.workB = sp->wB.ptr;
    u.dtype = dtype, u.backend = backend, u.fastmode = fastmode;
    return *status = HIPBLAS_STATUS_SUCCESS, true;
}
// counted front end (GEMMUL8_HOOK_STATS): true = the call was served here (emulated, or failed with *status set); false = native routine
bool try_emulate(int dtype, hipblasHandle_t handle, hipblasOperation_t ta, hipblasOperation_t tb, int m, int n, int k, const void* alpha,
                 const void* A, int lda, const void* B, int ldb, const void* beta, void* C, int ldc, hipblasStatus_t* status,
                 const hipStream_t* explicit_stream = nullptr) {
    if (tl_native_depth > 0) return false;  // inside a native pass-through of an outer hooked entry: the same call, already declined and counted
    const bool served = try_emulate_impl(dtype, handle, ta, tb, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, status, explicit_stream);
    count_call(served, dtype, m, n, k);
    return served;
}

// lt: the key is a hipblasLtHandle_t (hipblasLtDestroy) -- hipblasGetStream on such an object would be a type confusion
void release_state(hipblasHandle_t handle, bool lt = false) {
    std::shared_ptr<HandleState> sp;
    {
        std::lock_guard<std::mutex> g(g_map_mtx);
        auto it = g_map.find(handle);
        if (it == g_map.end()) return;
        sp = it->second;
        g_map.erase(it);
    }
    std::lock_guard<std::mutex> lk(sp->mtx);
    hipStream_t stream = nullptr;
    bool have = sp->have_stream;
    if (have) stream = sp->last_stream;
    else if (!lt && !sp->is_lt) {
        hipblasStatus_t st;
        stream = handle_stream(handle, &st);
        have = (st == HIPBLAS_STATUS_SUCCESS);
    }
    if (!have) (void)hipDeviceSynchronize();
    for (Buffer* b : {&sp->wA, &sp->wB, &sp->wC}) {
        if (!b->ptr) continue;
        hipError_t e = have ? hipFreeAsync(b->ptr, stream) : hipFree(b->ptr);
        if (e != hipSuccess && have && hipStreamSynchronize(stream) == hipSuccess) e = hipFree(b->ptr);
        if (e != hipSuccess) std::fprintf(stderr, "[GEMMUL8 HOOK] hipblasDestroy: freeing a workspace failed (%s)\n", hipGetErrorString(e));
        b->ptr = nullptr;
        b->size = 0;
    }
    if (sp->last_event) (void)hipEventDestroy(sp->last_event), sp->last_event = nullptr;
    for (BatchLane& ln : sp->lanes) {
        if (ln.w.ptr) {
            hipError_t e = ln.stream ? hipFreeAsync(ln.w.ptr, ln.stream) : hipFree(ln.w.ptr);
            if (e == hipSuccess && ln.stream) e = hipStreamSynchronize(ln.stream);
            if (e != hipSuccess) std::fprintf(stderr, "[GEMMUL8 HOOK] hipblasDestroy: freeing a batch workspace failed (%s)\n", hipGetErrorString(e));
            ln.w = Buffer{};
        }
        if (ln.done) (void)hipEventDestroy(ln.done), ln.done = nullptr;
        if (ln.stream) (void)hipStreamDestroy(ln.stream), ln.stream = nullptr;
    }
    if (sp->fork) (void)hipEventDestroy(sp->fork), sp->fork = nullptr;
}

#define OZ2_EARLY_OUT()                                           \
    if (m <= 0 || n <= 0 || k <= 0) return HIPBLAS_STATUS_SUCCESS; \
    if (!A || !B || !C) return HIPBLAS_STATUS_INVALID_VALUE;

}  // namespace

#pragma GCC visibility push(default)
extern "C" {

hipblasStatus_t hipblasDestroy(hipblasHandle_t handle) {
    release_state(handle);
    using Fn = hipblasStatus_t (*)(hipblasHandle_t);
    static Fn real = real_fn<Fn>("hipblasDestroy");
    NativeScope ns_; return real ? real(handle) : HIPBLAS_STATUS_NOT_INITIALIZED;
}

#define OZ2_GEMM_HOOK(NAME, T, CODE)                                                                                                   \
    hipblasStatus_t NAME(hipblasHandle_t handle, hipblasOperation_t transA, hipblasOperation_t transB, int m, int n, int k,             \
                         const T* alpha, const T* A, int lda, const T* B, int ldb, const T* beta, T* C, int ldc) {                      \
        OZ2_EARLY_OUT()                                                                                                                 \
        hipblasStatus_t st;                                                                                                             \
        if (try_emulate(CODE, handle, transA, transB, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, &st)) return st;                    \
        using Fn = hipblasStatus_t (*)(hipblasHandle_t, hipblasOperation_t, hipblasOperation_t, int, int, int, const T*, const T*, int, \
                                       const T*, int, const T*, T*, int);                                                               \
        static Fn real = real_fn<Fn>(#NAME);                                                                                            \
        NativeScope ns_; return real ? real(handle, transA, transB, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc) : HIPBLAS_STATUS_NOT_INITIALIZED;      \
    }
OZ2_GEMM_HOOK(hipblasSgemm, float, GEMMUL8_S)
OZ2_GEMM_HOOK(hipblasDgemm, double, GEMMUL8_D)
OZ2_GEMM_HOOK(hipblasCgemm, hipComplex, GEMMUL8_C)
OZ2_GEMM_HOOK(hipblasZgemm, hipDoubleComplex, GEMMUL8_Z)
#undef OZ2_GEMM_HOOK

hipblasStatus_t hipblasGemmEx(hipblasHandle_t handle, hipblasOperation_t transA, hipblasOperation_t transB, int m, int n, int k,
                              const void* alpha, const void* A, hipDataType aType, int lda, const void* B, hipDataType bType, int ldb,
                              const void* beta, void* C, hipDataType cType, int ldc, hipblasComputeType_t computeType,
                              hipblasGemmAlgo_t algo) {
    OZ2_EARLY_OUT()
    int dtype = -1;  // same (computeType, A/B/C type) dispatch as hook.cu:961-1030
    const bool same = (aType == bType && bType == cType);
    if (same && computeType == HIPBLAS_COMPUTE_32F && aType == HIP_R_32F) dtype = GEMMUL8_S;
    else if (same && computeType == HIPBLAS_COMPUTE_64F && aType == HIP_R_64F) dtype = GEMMUL8_D;
    else if (same && computeType == HIPBLAS_COMPUTE_32F && aType == HIP_C_32F) dtype = GEMMUL8_C;
    else if (same && computeType == HIPBLAS_COMPUTE_64F && aType == HIP_C_64F) dtype = GEMMUL8_Z;
    hipblasStatus_t st;
    if (dtype >= 0 && try_emulate(dtype, handle, transA, transB, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, &st)) return st;
    using Fn = hipblasStatus_t (*)(hipblasHandle_t, hipblasOperation_t, hipblasOperation_t, int, int, int, const void*, const void*,
                                   hipDataType, int, const void*, hipDataType, int, const void*, void*, hipDataType, int,
                                   hipblasComputeType_t, hipblasGemmAlgo_t);
    static Fn real = real_fn<Fn>("hipblasGemmEx");
    NativeScope ns_; return real ? real(handle, transA, transB, m, n, k, alpha, A, aType, lda, B, bType, ldb, beta, C, cType, ldc, computeType, algo)
                : HIPBLAS_STATUS_NOT_INITIALIZED;
}

// ROCm 7 also exports ILP64 twins (`_64`, int64_t dimensions) and hipblasGemmExWithFlags of the entry points above; the
// reference's CUDA-side hook predates them.  Same emulation when every dimension fits an int, the native routine otherwise.
static inline bool fits_int(int64_t a, int64_t b, int64_t c, int64_t d, int64_t e, int64_t f) {
    const int64_t lim = 2147483647;
    return a <= lim && b <= lim && c <= lim && d <= lim && e <= lim && f <= lim;
}
#define OZ2_GEMM_HOOK_64(NAME, T, CODE)                                                                                                  \
    hipblasStatus_t NAME(hipblasHandle_t handle, hipblasOperation_t transA, hipblasOperation_t transB, int64_t m, int64_t n, int64_t k,   \
                         const T* alpha, const T* A, int64_t lda, const T* B, int64_t ldb, const T* beta, T* C, int64_t ldc) {            \
        OZ2_EARLY_OUT()                                                                                                                   \
        hipblasStatus_t st;                                                                                                               \
        if (fits_int(m, n, k, lda, ldb, ldc) &&                                                                                           \
            try_emulate(CODE, handle, transA, transB, (int)m, (int)n, (int)k, alpha, A, (int)lda, B, (int)ldb, beta, C, (int)ldc, &st))   \
            return st;                                                                                                                    \
        using Fn = hipblasStatus_t (*)(hipblasHandle_t, hipblasOperation_t, hipblasOperation_t, int64_t, int64_t, int64_t, const T*,      \
                                       const T*, int64_t, const T*, int64_t, const T*, T*, int64_t);                                      \
        static Fn real = real_fn<Fn>(#NAME);                                                                                              \
        NativeScope ns_; return real ? real(handle, transA, transB, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc) : HIPBLAS_STATUS_NOT_INITIALIZED;        \
    }
OZ2_GEMM_HOOK_64(hipblasSgemm_64, float, GEMMUL8_S)
OZ2_GEMM_HOOK_64(hipblasDgemm_64, double, GEMMUL8_D)
OZ2_GEMM_HOOK_64(hipblasCgemm_64, hipComplex, GEMMUL8_C)
OZ2_GEMM_HOOK_64(hipblasZgemm_64, hipDoubleComplex, GEMMUL8_Z)
#undef OZ2_GEMM_HOOK_64

static int gemm_ex_dtype(hipDataType aType, hipDataType bType, hipDataType cType, hipblasComputeType_t computeType) {
    const bool same = (aType == bType && bType == cType);  // same (computeType, A/B/C type) dispatch as hook.cu:961-1030
    if (same && computeType == HIPBLAS_COMPUTE_32F && aType == HIP_R_32F) return GEMMUL8_S;
    if (same && computeType == HIPBLAS_COMPUTE_64F && aType == HIP_R_64F) return GEMMUL8_D;
    if (same && computeType == HIPBLAS_COMPUTE_32F && aType == HIP_C_32F) return GEMMUL8_C;
    if (same && computeType == HIPBLAS_COMPUTE_64F && aType == HIP_C_64F) return GEMMUL8_Z;
    return -1;
}

hipblasStatus_t hipblasGemmExWithFlags(hipblasHandle_t handle, hipblasOperation_t transA, hipblasOperation_t transB, int m, int n, int k,
                                       const void* alpha, const void* A, hipDataType aType, int lda, const void* B, hipDataType bType, int ldb,
                                       const void* beta, void* C, hipDataType cType, int ldc, hipblasComputeType_t computeType,
                                       hipblasGemmAlgo_t algo, hipblasGemmFlags_t flags) {
    OZ2_EARLY_OUT()
    const int dtype = gemm_ex_dtype(aType, bType, cType, computeType);
    hipblasStatus_t st;
    if (dtype >= 0 && try_emulate(dtype, handle, transA, transB, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, &st)) return st;
    using Fn = hipblasStatus_t (*)(hipblasHandle_t, hipblasOperation_t, hipblasOperation_t, int, int, int, const void*, const void*,
                                   hipDataType, int, const void*, hipDataType, int, const void*, void*, hipDataType, int,
                                   hipblasComputeType_t, hipblasGemmAlgo_t, hipblasGemmFlags_t);
    static Fn real = real_fn<Fn>("hipblasGemmExWithFlags");
    NativeScope ns_; return real ? real(handle, transA, transB, m, n, k, alpha, A, aType, lda, B, bType, ldb, beta, C, cType, ldc, computeType, algo, flags)
                : HIPBLAS_STATUS_NOT_INITIALIZED;
}

hipblasStatus_t hipblasGemmEx_64(hipblasHandle_t handle, hipblasOperation_t transA, hipblasOperation_t transB, int64_t m, int64_t n, int64_t k,
                                 const void* alpha, const void* A, hipDataType aType, int64_t lda, const void* B, hipDataType bType,
                                 int64_t ldb, const void* beta, void* C, hipDataType cType, int64_t ldc, hipblasComputeType_t computeType,
                                 hipblasGemmAlgo_t algo) {
    OZ2_EARLY_OUT()
    const int dtype = gemm_ex_dtype(aType, bType, cType, computeType);
    hipblasStatus_t st;
    if (dtype >= 0 && fits_int(m, n, k, lda, ldb, ldc) &&
        try_emulate(dtype, handle, transA, transB, (int)m, (int)n, (int)k, alpha, A, (int)lda, B, (int)ldb, beta, C, (int)ldc, &st))
        return st;
    using Fn = hipblasStatus_t (*)(hipblasHandle_t, hipblasOperation_t, hipblasOperation_t, int64_t, int64_t, int64_t, const void*,
                                   const void*, hipDataType, int64_t, const void*, hipDataType, int64_t, const void*, void*, hipDataType,
                                   int64_t, hipblasComputeType_t, hipblasGemmAlgo_t);
    static Fn real = real_fn<Fn>("hipblasGemmEx_64");
    NativeScope ns_; return real ? real(handle, transA, transB, m, n, k, alpha, A, aType, lda, B, bType, ldb, beta, C, cType, ldc, computeType, algo)
                : HIPBLAS_STATUS_NOT_INITIALIZED;
}

hipblasStatus_t hipblasGemmExWithFlags_64(hipblasHandle_t handle, hipblasOperation_t transA, hipblasOperation_t transB, int64_t m, int64_t n,
                                          int64_t k, const void* alpha, const void* A, hipDataType aType, int64_t lda, const void* B,
                                          hipDataType bType, int64_t ldb, const void* beta, void* C, hipDataType cType, int64_t ldc,
                                          hipblasComputeType_t computeType, hipblasGemmAlgo_t algo, hipblasGemmFlags_t flags) {
    OZ2_EARLY_OUT()
    const int dtype = gemm_ex_dtype(aType, bType, cType, computeType);
    hipblasStatus_t st;
    if (dtype >= 0 && fits_int(m, n, k, lda, ldb, ldc) &&
        try_emulate(dtype, handle, transA, transB, (int)m, (int)n, (int)k, alpha, A, (int)lda, B, (int)ldb, beta, C, (int)ldc, &st))
        return st;
    using Fn = hipblasStatus_t (*)(hipblasHandle_t, hipblasOperation_t, hipblasOperation_t, int64_t, int64_t, int64_t, const void*,
                                   const void*, hipDataType, int64_t, const void*, hipDataType, int64_t, const void*, void*, hipDataType,
                                   int64_t, hipblasComputeType_t, hipblasGemmAlgo_t, hipblasGemmFlags_t);
    static Fn real = real_fn<Fn>("hipblasGemmExWithFlags_64");
    NativeScope ns_; return real ? real(handle, transA, transB, m, n, k, alpha, A, aType, lda, B, bType, ldb, beta, C, cType, ldc, computeType, algo, flags)
                : HIPBLAS_STATUS_NOT_INITIALIZED;
}

// Strided-batched entry points (not hooked by the reference; PyTorch's bmm uses them).  alpha/beta are shared by the batch; element
// strides are in units of the matrix type.  The items of a batch are independent, and below ~2048^3 one emulated GEMM is ten
// latency-bound launches that leave most of the chip idle.  Default: one set of launches for the whole batch (below).  Otherwise
// (GEMMUL8_BATCH_FUSED=0, GEMMUL8_DIST) the batch is spread over GEMMUL8_BATCH_STREAMS lanes (default 4, 1 =
// serial loop on the handle's stream): lane 0 is the handle's stream with the handle's buffers, every other lane has its own
// non-blocking stream and workspace; the lanes fork from the handle's stream with an event and join it again before the call
// returns, so the call stays stream-ordered for the application (and capturable in a HIP graph after one warm-up call).
static bool emulate_batch(int dtype, size_t elem, hipblasHandle_t handle, hipblasOperation_t ta, hipblasOperation_t tb, int m, int n, int k,
                          const void* alpha, const void* A, int lda, long long sa, const void* B, int ldb, long long sb, const void* beta,
                          void* C, int ldc, long long sc, int batch, hipblasStatus_t* status, const hipStream_t* explicit_stream = nullptr) {
    *status = HIPBLAS_STATUS_SUCCESS;
    if (tl_native_depth > 0) return false;  // see try_emulate
    // First choice (INT8 backend, GEMMUL8_BATCH_FUSED != 0): the whole batch as ONE set of launches (gemmul8_gemm_batched: the items in
    // gridDim.z of every kernel) -- a batch of small matrices then fills the chip and costs ten launches, not ten per item.
    if (batch > 1 && env_u64("GEMMUL8_BATCH_FUSED", 1) != 0 && dist_kind_from_env() < 0) {
        const TypeInfo& ti = kTypes[dtype];
        const unsigned N = (unsigned)env_u64(ti.nmod, 0);
        if (N < 2u || N > ti.max_moduli) return false;
        const int backend = env_backend("GEMMUL8_BACKEND", 0, false);
        if (below_floor(dtype, m, n, k, N, env_one(ti.fast), backend, (double)batch)) return false;
        if (k <= (backend == GEMMUL8_FP8 ? 65536 : (1 << 17))) {
            const bool fastmode = env_one(ti.fast);
            auto sp = state_of(handle);
            std::lock_guard<std::mutex> lk(sp->mtx);
            if (explicit_stream) sp->is_lt = true;
            init_max_workspace();
            hipblasStatus_t st = HIPBLAS_STATUS_SUCCESS;
            hipStream_t stream = explicit_stream ? *explicit_stream : handle_stream(handle, &st);
            if (st != HIPBLAS_STATUS_SUCCESS) return *status = st, true;
            if ((st = order_streams(*sp, stream)) != HIPBLAS_STATUS_SUCCESS) return *status = st, true;
            // the items' workspaces are consecutive: bound the buffer (GEMMUL8_BATCH_WORKSPACE_MB, default 4096) and run the batch in
            // chunks of as many items as fit; if even that cannot be allocated the per-item path below takes over
            const size_t item = gemmul8_work_size_batched(ti.cplx, backend, (size_t)m, (size_t)n, (size_t)k, N, 1) - 256;
            const size_t budget = (size_t)env_u64("GEMMUL8_BATCH_WORKSPACE_MB", 4096) << 20;
            const size_t per_chunk = std::max<size_t>(1, std::min<size_t>((size_t)batch, item ? budget / item : (size_t)batch));
            if (grow(sp->wC, item * per_chunk + 256, stream, "workC (batched)") == HIPBLAS_STATUS_SUCCESS) {
                sp->last.valid = false;  // the skip-scaling cache describes single calls; the planes are overwritten here
                int rc = 0;
                for (size_t b0 = 0; b0 < (size_t)batch && rc == 0; b0 += per_chunk) {
                    const size_t nb = std::min(per_chunk, (size_t)batch - b0);
                    rc = gemmul8_gemm_batched(stream, dtype, backend, (int)ta, (int)tb, (size_t)m, (size_t)n, (size_t)k, alpha,
                                              (const char*)A + (long long)b0 * sa * (long long)elem, (size_t)lda, sa,
                                              (const char*)B + (long long)b0 * sb * (long long)elem, (size_t)ldb, sb, beta,
                                              (char*)C + (long long)b0 * sc * (long long)elem, (size_t)ldc, sc, nb, N, fastmode, sp->wC.ptr);
                    if (rc < 0 && b0 > 0) rc = 1;  // declined after earlier chunks were written: cannot hand the call to another path
                }
                if (rc == 0) return count_call(true, dtype, m, n, k, (double)batch), true;
                if (rc > 0) return *status = HIPBLAS_STATUS_INTERNAL_ERROR, true;
                // negative on the first chunk: declined (nothing written) -- fall through to the per-item path
            } else {
                (void)hipGetLastError();  // allocation failed: clear the sticky error, use the per-item path
            }
        }
    }
    auto item = [&](int b, const void** a, const void** bb, void** c) {
        *a = (const char*)A + (size_t)b * sa * elem, *bb = (const char*)B + (size_t)b * sb * elem, *c = (char*)C + (size_t)b * sc * elem;
    };
    const void *Ai, *Bi;
    void* Ci;
    hipblasStatus_t st;
    // item 0 on the handle's stream decides whether the environment selects emulation for this call at all
    item(0, &Ai, &Bi, &Ci);
    if (!try_emulate(dtype, handle, ta, tb, m, n, k, alpha, Ai, lda, Bi, ldb, beta, Ci, ldc, &st, explicit_stream)) return false;
    if (st != HIPBLAS_STATUS_SUCCESS) return *status = st, true;
    int lanes = (int)env_u64("GEMMUL8_BATCH_STREAMS", 4);
    lanes = std::max(1, std::min({lanes, kMaxBatchLanes, batch}));
    const TypeInfo& ti = kTypes[dtype];
    const unsigned N = (unsigned)env_u64(ti.nmod, 0);
    const bool fastmode = env_one(ti.fast);
    const int backend = env_backend("GEMMUL8_BACKEND", 0, false);
    if (lanes > 1 && dist_kind_from_env() >= 0) lanes = 1;  // the sharded path keeps its collectives on one stream
    auto sp = state_of(handle);
    hipStream_t main_stream = nullptr;
    if (lanes > 1) {
        std::lock_guard<std::mutex> lk(sp->mtx);
        main_stream = sp->last_stream;  // set by item 0
        bool ok = sp->fork || hipEventCreateWithFlags(&sp->fork, hipEventDisableTiming) == hipSuccess;
        const size_t need = gemmul8_work_size(ti.cplx, backend, (size_t)m, (size_t)n, (size_t)k, N, 0, 0, nullptr, nullptr);
        ok = ok && hipEventRecord(sp->fork, main_stream) == hipSuccess;
        for (int l = 1; l < lanes && ok; ++l) {
            BatchLane& ln = sp->lanes[l];
            if (!ln.stream) ok = hipStreamCreateWithFlags(&ln.stream, hipStreamNonBlocking) == hipSuccess;
            if (ok && !ln.done) ok = hipEventCreateWithFlags(&ln.done, hipEventDisableTiming) == hipSuccess;
            ok = ok && hipStreamWaitEvent(ln.stream, sp->fork, 0) == hipSuccess;
            ok = ok && grow(ln.w, need, ln.stream, "batch lane") == HIPBLAS_STATUS_SUCCESS;
        }
        if (!ok) lanes = 1;  // fall back to the serial loop
    }
    for (int b = 1; b < batch; ++b) {
        item(b, &Ai, &Bi, &Ci);
        const int l = b % lanes;
        if (l == 0) {
            const bool done = try_emulate(dtype, handle, ta, tb, m, n, k, alpha, Ai, lda, Bi, ldb, beta, Ci, ldc, &st, explicit_stream);
            if (!done || st != HIPBLAS_STATUS_SUCCESS) {  // stop issuing items; the forked lanes are still joined below
                *status = done ? st : HIPBLAS_STATUS_INTERNAL_ERROR;
                break;
            }
        } else {
            std::lock_guard<std::mutex> lk(sp->mtx);
            BatchLane& ln = sp->lanes[l];
            const int rc = gemmul8_gemm(ln.stream, dtype, backend, (int)ta, (int)tb, (size_t)m, (size_t)n, (size_t)k, alpha, Ai, (size_t)lda, Bi,
                                        (size_t)ldb, beta, Ci, (size_t)ldc, N, fastmode, ln.w.ptr, nullptr, nullptr, 0, 0, 0, 0, nullptr);
            if (rc != 0) *status = HIPBLAS_STATUS_INTERNAL_ERROR;  // item 0 ran with the same shape and switches: should not happen
        }
    }
    if (lanes > 1) {  // join: the handle's stream waits for every lane
        std::lock_guard<std::mutex> lk(sp->mtx);
        for (int l = 1; l < lanes; ++l) {
            BatchLane& ln = sp->lanes[l];
            if (hipEventRecord(ln.done, ln.stream) != hipSuccess || hipStreamWaitEvent(main_stream, ln.done, 0) != hipSuccess)
                *status = HIPBLAS_STATUS_INTERNAL_ERROR;
        }
    }
    return true;
}

#define OZ2_SB_HOOK(NAME, T, CODE)                                                                                                       \
    hipblasStatus_t NAME(hipblasHandle_t handle, hipblasOperation_t transA, hipblasOperation_t transB, int m, int n, int k,               \
                         const T* alpha, const T* A, int lda, long long strideA, const T* B, int ldb, long long strideB, const T* beta,  \
                         T* C, int ldc, long long strideC, int batchCount) {                                                             \
        hipblasStatus_t st;                                                                                                               \
        if (m > 0 && n > 0 && k > 0 && batchCount > 0 && A && B && C && alpha && beta &&                                                  \
            emulate_batch(CODE, sizeof(T), handle, transA, transB, m, n, k, alpha, A, lda, strideA, B, ldb, strideB, beta, C, ldc,        \
                          strideC, batchCount, &st))                                                                                      \
            return st;                                                                                                                    \
        using Fn = hipblasStatus_t (*)(hipblasHandle_t, hipblasOperation_t, hipblasOperation_t, int, int, int, const T*, const T*, int,   \
                                       long long, const T*, int, long long, const T*, T*, int, long long, int);                          \
        static Fn real = real_fn<Fn>(#NAME);                                                                                              \
        NativeScope ns_; return real ? real(handle, transA, transB, m, n, k, alpha, A, lda, strideA, B, ldb, strideB, beta, C, ldc, strideC, batchCount)   \
                    : HIPBLAS_STATUS_NOT_INITIALIZED;                                                                                     \
    }
OZ2_SB_HOOK(hipblasSgemmStridedBatched, float, GEMMUL8_S)
OZ2_SB_HOOK(hipblasDgemmStridedBatched, double, GEMMUL8_D)
OZ2_SB_HOOK(hipblasCgemmStridedBatched, hipComplex, GEMMUL8_C)
OZ2_SB_HOOK(hipblasZgemmStridedBatched, hipDoubleComplex, GEMMUL8_Z)
#undef OZ2_SB_HOOK

hipblasStatus_t hipblasGemmStridedBatchedEx(hipblasHandle_t handle, hipblasOperation_t transA, hipblasOperation_t transB, int m, int n, int k,
                                            const void* alpha, const void* A, hipDataType aType, int lda, hipblasStride strideA, const void* B,
                                            hipDataType bType, int ldb, hipblasStride strideB, const void* beta, void* C, hipDataType cType,
                                            int ldc, hipblasStride strideC, int batchCount, hipblasComputeType_t computeType,
                                            hipblasGemmAlgo_t algo) {
    int dtype = -1;
    size_t elem = 0;
    const bool same = (aType == bType && bType == cType);
    if (same && computeType == HIPBLAS_COMPUTE_32F && aType == HIP_R_32F) dtype = GEMMUL8_S, elem = 4;
    else if (same && computeType == HIPBLAS_COMPUTE_64F && aType == HIP_R_64F) dtype = GEMMUL8_D, elem = 8;
    else if (same && computeType == HIPBLAS_COMPUTE_32F && aType == HIP_C_32F) dtype = GEMMUL8_C, elem = 8;
    else if (same && computeType == HIPBLAS_COMPUTE_64F && aType == HIP_C_64F) dtype = GEMMUL8_Z, elem = 16;
    hipblasStatus_t st;
    if (dtype >= 0 && m > 0 && n > 0 && k > 0 && batchCount > 0 && A && B && C && alpha && beta &&
        emulate_batch(dtype, elem, handle, transA, transB, m, n, k, alpha, A, lda, (long long)strideA, B, ldb, (long long)strideB, beta, C, ldc,
                      (long long)strideC, batchCount, &st))
        return st;
    using Fn = hipblasStatus_t (*)(hipblasHandle_t, hipblasOperation_t, hipblasOperation_t, int, int, int, const void*, const void*,
                                   hipDataType, int, hipblasStride, const void*, hipDataType, int, hipblasStride, const void*, void*,
                                   hipDataType, int, hipblasStride, int, hipblasComputeType_t, hipblasGemmAlgo_t);
    static Fn real = real_fn<Fn>("hipblasGemmStridedBatchedEx");
    NativeScope ns_; return real ? real(handle, transA, transB, m, n, k, alpha, A, aType, lda, strideA, B, bType, ldb, strideB, beta, C, cType, ldc, strideC,
                       batchCount, computeType, algo)
                : HIPBLAS_STATUS_NOT_INITIALIZED;
}


// ---- rocBLAS entry points, opt-in with GEMMUL8_HOOK_ROCBLAS=1 (no counterpart in the reference: src/hook.cu:846-1055 hooks the cuBLAS / hipBLAS
// names only).  For applications that call rocBLAS directly.  The handle of a hipBLAS call IS the rocBLAS handle, so a call the hipBLAS hooks
// above declined arrives here again through the real hipBLAS and is declined again by the same rules.  rocSOLVER / hipSOLVER factorizations do
// NOT come through here: they call rocBLAS's internal C++ templates, not these exported C entry points (INTEGRATION.md "What the hook
// reaches", tests/test_gpu_hook_reach.py).  rocblas_operation / rocblas_status are plain ints here (111 / 112 / 113 = the hipBLAS values;
// 0 = success, 6 = internal error); rocblas_int is 32-bit in this build of rocBLAS (rocblas-types.h:79).
}  // extern "C"
#pragma GCC visibility pop
namespace {
void* mapped_rocblas() {
    static void* h = mapped_library("librocblas.so");
    return h;
}
template <typename Fn> Fn real_rocblas(const char* name) {
    void* f = dlsym(RTLD_NEXT, name);
    if (!f)
        if (void* h = mapped_rocblas()) f = dlsym(h, name);
    return reinterpret_cast<Fn>(f);
}
bool rocblas_stream(void* handle, hipStream_t* s) {
    using Fn = int (*)(void*, hipStream_t*);
    static Fn fn = real_rocblas<Fn>("rocblas_get_stream");
    return fn && fn(handle, s) == 0;
}
int rocblas_status_of(hipblasStatus_t st) { return st == HIPBLAS_STATUS_SUCCESS ? 0 : st == HIPBLAS_STATUS_ALLOC_FAILED ? 5 : 6; }

// rocblas_internal_gemm_template is an INTERNAL, unversioned C++ symbol: the mangled name pins the parameter TYPES but not their meaning
// (offsets in elements, strides, the batch count), and a ROCm point release may change those without changing the name -- silent argument
// corruption instead of a clean pass-through.  The interposition is therefore limited to the rocBLAS releases it was run against
// (tests/test_gpu_hook_reach.py on PyTorch's bundled 5.0.x, tests/cpp/test_hook_rocblas.cpp on ROCm 7.2's 5.2.x); any other version string
// -- or a rocBLAS without rocblas_get_version_string -- takes the pass-through, with one log line.  GEMMUL8_ROCBLAS_ABI_UNCHECKED=1 overrides.
constexpr const char* kTestedRocblas[] = {"5.0.", "5.2."};
bool rocblas_version_tested(const char* v) {
    if (!v) return false;
    for (const char* pre : kTestedRocblas)
        if (std::strncmp(v, pre, std::strlen(pre)) == 0) return true;
    return false;
}
bool rocblas_internal_abi_ok() {
    static const bool ok = [] {
        if (env_one("GEMMUL8_ROCBLAS_ABI_UNCHECKED")) return true;
        using SizeFn = int (*)(size_t*);
        using StrFn = int (*)(char*, size_t);
        SizeFn fsz = real_rocblas<SizeFn>("rocblas_get_version_string_size");
        StrFn fstr = real_rocblas<StrFn>("rocblas_get_version_string");
        char buf[128] = "";
        size_t len = 0;
        const bool have = fsz && fstr && fsz(&len) == 0 && len > 0 && len <= sizeof(buf) && fstr(buf, len) == 0;
        const bool good = have && rocblas_version_tested(buf);
        if (!good && env_one("GEMMUL8_HOOK_ROCBLAS"))
            std::fprintf(stderr, "[GEMMUL8 HOOK] rocBLAS version '%s' is not one this build was tested with (5.0.x, 5.2.x): rocblas_internal_gemm_template "
                                 "(rocSOLVER's trailing updates) is NOT intercepted; the exported rocblas_*gemm entry points still are "
                                 "(GEMMUL8_ROCBLAS_ABI_UNCHECKED=1 overrides)\n", have ? buf : "unknown");
        return good;
    }();
    return ok;
}
}  // namespace
extern "C" GEMMUL8_API int gemmul8_hook_rocblas_version_tested(const char* version) { return rocblas_version_tested(version) ? 1 : 0; }
#pragma GCC visibility push(default)
extern "C" {

int rocblas_destroy_handle(void* handle) {
    if (handle) release_state((hipblasHandle_t)handle, true);
    using Fn = int (*)(void*);
    static Fn real = real_rocblas<Fn>("rocblas_destroy_handle");
    NativeScope ns_; return real ? real(handle) : 6;
}

#define OZ2_ROCBLAS_GEMM_HOOK(NAME, T, CODE)                                                                                             \
    int NAME(void* handle, int transA, int transB, int m, int n, int k, const T* alpha, const T* A, int lda, const T* B, int ldb,        \
             const T* beta, T* C, int ldc) {                                                                                             \
        using Fn = int (*)(void*, int, int, int, int, int, const T*, const T*, int, const T*, int, const T*, T*, int);                   \
        static Fn real = real_rocblas<Fn>(#NAME);                                                                                        \
        hipStream_t s_;                                                                                                                  \
        hipblasStatus_t st_;                                                                                                             \
        if (env_one("GEMMUL8_HOOK_ROCBLAS") && handle && m > 0 && n > 0 && k > 0 && alpha && A && B && beta && C &&                      \
            transA >= 111 && transA <= 113 && transB >= 111 && transB <= 113 && rocblas_stream(handle, &s_) &&                           \
            try_emulate(CODE, (hipblasHandle_t)handle, (hipblasOperation_t)transA, (hipblasOperation_t)transB, m, n, k, alpha, A, lda, B, \
                        ldb, beta, C, ldc, &st_, &s_))                                                                                   \
            return rocblas_status_of(st_);                                                                                               \
        NativeScope ns_; return real ? real(handle, transA, transB, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc) : 6;                                    \
    }
OZ2_ROCBLAS_GEMM_HOOK(rocblas_sgemm, float, GEMMUL8_S)
OZ2_ROCBLAS_GEMM_HOOK(rocblas_dgemm, double, GEMMUL8_D)
OZ2_ROCBLAS_GEMM_HOOK(rocblas_cgemm, hipComplex, GEMMUL8_C)
OZ2_ROCBLAS_GEMM_HOOK(rocblas_zgemm, hipDoubleComplex, GEMMUL8_Z)
#undef OZ2_ROCBLAS_GEMM_HOOK

#define OZ2_ROCBLAS_SB_HOOK(NAME, T, CODE)                                                                                               \
    int NAME(void* handle, int transA, int transB, int m, int n, int k, const T* alpha, const T* A, int lda, long long strideA,          \
             const T* B, int ldb, long long strideB, const T* beta, T* C, int ldc, long long strideC, int batchCount) {                  \
        using Fn = int (*)(void*, int, int, int, int, int, const T*, const T*, int, long long, const T*, int, long long, const T*, T*,   \
                           int, long long, int);                                                                                         \
        static Fn real = real_rocblas<Fn>(#NAME);                                                                                        \
        hipStream_t s_;                                                                                                                  \
        hipblasStatus_t st_;                                                                                                             \
        if (env_one("GEMMUL8_HOOK_ROCBLAS") && handle && m > 0 && n > 0 && k > 0 && batchCount > 0 && alpha && A && B && beta && C &&    \
            transA >= 111 && transA <= 113 && transB >= 111 && transB <= 113 && rocblas_stream(handle, &s_) &&                           \
            emulate_batch(CODE, sizeof(T), (hipblasHandle_t)handle, (hipblasOperation_t)transA, (hipblasOperation_t)transB, m, n, k,     \
                          alpha, A, lda, strideA, B, ldb, strideB, beta, C, ldc, strideC, batchCount, &st_, &s_))                        \
            return rocblas_status_of(st_);                                                                                               \
        NativeScope ns_; return real ? real(handle, transA, transB, m, n, k, alpha, A, lda, strideA, B, ldb, strideB, beta, C, ldc, strideC, batchCount) : 6; \
    }
OZ2_ROCBLAS_SB_HOOK(rocblas_sgemm_strided_batched, float, GEMMUL8_S)
OZ2_ROCBLAS_SB_HOOK(rocblas_dgemm_strided_batched, double, GEMMUL8_D)
OZ2_ROCBLAS_SB_HOOK(rocblas_cgemm_strided_batched, hipComplex, GEMMUL8_C)
OZ2_ROCBLAS_SB_HOOK(rocblas_zgemm_strided_batched, hipDoubleComplex, GEMMUL8_Z)
#undef OZ2_ROCBLAS_SB_HOOK

// rocSOLVER / hipSOLVER factorizations (getrf, geqrf, potrf ...: what torch.linalg.lu_factor / solve / qr run on) do not call the C entry
// points above: their trailing updates go through rocBLAS's exported C++ template rocblas_internal_gemm_template<T> (and its _64 form).
// Those ARE dynamic symbols, so the same opt-in interposes them -- by their mangled names, which belong to THIS rocBLAS ABI (ROCm 7.2:
// nm -D librocsolver.so | grep internal_gemm); if a later rocBLAS changes the signature the names no longer match, nothing is intercepted,
// and tests/test_gpu_hook_reach.py says so.  Strided batch (batch_count > 1) and element offsets as rocBLAS defines them.
}  // extern "C"
template <typename T, typename I>
static int rocblas_internal_gemm_hook(const char* mangled, int code, void* handle, int transA, int transB, I m, I n, I k, const T* alpha, const T* A,
                                      long offA, I lda, long strideA, const T* B, long offB, I ldb, long strideB, const T* beta, T* C, long offC,
                                      I ldc, long strideC, I batch) {
    using Fn = int (*)(void*, int, int, I, I, I, const T*, const T*, long, I, long, const T*, long, I, long, const T*, T*, long, I, long, I);
    static Fn real = real_rocblas<Fn>(mangled);  // one symbol per <T, I> instantiation: resolved once
    hipStream_t s_;
    hipblasStatus_t st_;
    const bool fits = m > 0 && n > 0 && k > 0 && batch > 0 && (long long)m <= 2147483647 && (long long)n <= 2147483647 && (long long)k <= 2147483647 &&
                      (long long)lda <= 2147483647 && (long long)ldb <= 2147483647 && (long long)ldc <= 2147483647 && (long long)batch <= 2147483647;
    if (fits && env_one("GEMMUL8_HOOK_ROCBLAS") && rocblas_internal_abi_ok() && handle && alpha && A && B && beta && C && transA >= 111 && transA <= 113 &&
        transB >= 111 && transB <= 113 && rocblas_stream(handle, &s_)) {
        const T *Ao = A + offA, *Bo = B + offB;
        T* Co = C + offC;
        const bool served = batch == 1 ? try_emulate(code, (hipblasHandle_t)handle, (hipblasOperation_t)transA, (hipblasOperation_t)transB, (int)m, (int)n,
                                                     (int)k, alpha, Ao, (int)lda, Bo, (int)ldb, beta, Co, (int)ldc, &st_, &s_)
                                       : emulate_batch(code, sizeof(T), (hipblasHandle_t)handle, (hipblasOperation_t)transA, (hipblasOperation_t)transB,
                                                       (int)m, (int)n, (int)k, alpha, Ao, (int)lda, strideA, Bo, (int)ldb, strideB, beta, Co, (int)ldc,
                                                       strideC, (int)batch, &st_, &s_);
        if (served) return rocblas_status_of(st_);
    }
    NativeScope ns_; return real ? real(handle, transA, transB, m, n, k, alpha, A, offA, lda, strideA, B, offB, ldb, strideB, beta, C, offC, ldc, strideC, batch) : 6;
}
extern "C" {
#define OZ2_ROCBLAS_INTERNAL(FN, T, CODE, SYM32, SYM64)                                                                                   \
    int FN##_32(void* h, int ta, int tb, int m, int n, int k, const T* al, const T* A, long oa, int lda, long sa, const T* B, long ob, int ldb, \
                long sb, const T* be, T* C, long oc, int ldc, long sc, int bc) __asm__(SYM32);                                            \
    int FN##_32(void* h, int ta, int tb, int m, int n, int k, const T* al, const T* A, long oa, int lda, long sa, const T* B, long ob, int ldb, \
                long sb, const T* be, T* C, long oc, int ldc, long sc, int bc) {                                                          \
        return rocblas_internal_gemm_hook<T, int>(SYM32, CODE, h, ta, tb, m, n, k, al, A, oa, lda, sa, B, ob, ldb, sb, be, C, oc, ldc, sc, bc); \
    }                                                                                                                                     \
    int FN##_64(void* h, int ta, int tb, long m, long n, long k, const T* al, const T* A, long oa, long lda, long sa, const T* B, long ob,  \
                long ldb, long sb, const T* be, T* C, long oc, long ldc, long sc, long bc) __asm__(SYM64);                                \
    int FN##_64(void* h, int ta, int tb, long m, long n, long k, const T* al, const T* A, long oa, long lda, long sa, const T* B, long ob,  \
                long ldb, long sb, const T* be, T* C, long oc, long ldc, long sc, long bc) {                                              \
        return rocblas_internal_gemm_hook<T, long>(SYM64, CODE, h, ta, tb, m, n, k, al, A, oa, lda, sa, B, ob, ldb, sb, be, C, oc, ldc, sc, bc); \
    }
OZ2_ROCBLAS_INTERNAL(oz2_rb_int_gemm_s, float, GEMMUL8_S,
                     "_Z30rocblas_internal_gemm_templateIfE15rocblas_status_P15_rocblas_handle18rocblas_operation_S3_iiiPKT_S6_lilS6_lilS6_PS4_lili",
                     "_Z33rocblas_internal_gemm_template_64IfE15rocblas_status_P15_rocblas_handle18rocblas_operation_S3_lllPKT_S6_lllS6_lllS6_PS4_llll")
OZ2_ROCBLAS_INTERNAL(oz2_rb_int_gemm_d, double, GEMMUL8_D,
                     "_Z30rocblas_internal_gemm_templateIdE15rocblas_status_P15_rocblas_handle18rocblas_operation_S3_iiiPKT_S6_lilS6_lilS6_PS4_lili",
                     "_Z33rocblas_internal_gemm_template_64IdE15rocblas_status_P15_rocblas_handle18rocblas_operation_S3_lllPKT_S6_lllS6_lllS6_PS4_llll")
OZ2_ROCBLAS_INTERNAL(oz2_rb_int_gemm_c, hipComplex, GEMMUL8_C,
                     "_Z30rocblas_internal_gemm_templateI19rocblas_complex_numIfEE15rocblas_status_P15_rocblas_handle18rocblas_operation_S5_iiiPKT_S8_lilS8_lilS8_PS6_lili",
                     "_Z33rocblas_internal_gemm_template_64I19rocblas_complex_numIfEE15rocblas_status_P15_rocblas_handle18rocblas_operation_S5_lllPKT_S8_lllS8_lllS8_PS6_llll")
OZ2_ROCBLAS_INTERNAL(oz2_rb_int_gemm_z, hipDoubleComplex, GEMMUL8_Z,
                     "_Z30rocblas_internal_gemm_templateI19rocblas_complex_numIdEE15rocblas_status_P15_rocblas_handle18rocblas_operation_S5_iiiPKT_S8_lilS8_lilS8_PS6_lili",
                     "_Z33rocblas_internal_gemm_template_64I19rocblas_complex_numIdEE15rocblas_status_P15_rocblas_handle18rocblas_operation_S5_lllPKT_S8_lllS8_lllS8_PS6_llll")
#undef OZ2_ROCBLAS_INTERNAL

// rocblas_gemm_ex: D = alpha op(A) op(B) + beta C.  Emulated for the four plain types (all of a / b / c / d / compute the same type) when it
// is the in-place form (c == d, ldc == ldd) -- what rocBLAS's own clients and hipBLAS's GemmEx issue; anything else goes to rocBLAS.
int rocblas_gemm_ex(void* handle, int transA, int transB, int m, int n, int k, const void* alpha, const void* a, int a_type, int lda,
                    const void* b, int b_type, int ldb, const void* beta, const void* c, int c_type, int ldc, void* d, int d_type, int ldd,
                    int compute_type, int algo, int32_t solution_index, uint32_t flags) {
    using Fn = int (*)(void*, int, int, int, int, int, const void*, const void*, int, int, const void*, int, int, const void*, const void*, int,
                       int, void*, int, int, int, int, int32_t, uint32_t);
    static Fn real = real_rocblas<Fn>("rocblas_gemm_ex");
    const bool same = a_type == b_type && b_type == c_type && c_type == d_type && d_type == compute_type;
    const int dtype = !same ? -1 : a_type == 151 ? GEMMUL8_S : a_type == 152 ? GEMMUL8_D : a_type == 154 ? GEMMUL8_C : a_type == 155 ? GEMMUL8_Z : -1;
    hipStream_t s_;
    hipblasStatus_t st_;
    if (dtype >= 0 && env_one("GEMMUL8_HOOK_ROCBLAS") && handle && m > 0 && n > 0 && k > 0 && alpha && a && b && beta && d && c == d &&
        ldc == ldd && transA >= 111 && transA <= 113 && transB >= 111 && transB <= 113 && rocblas_stream(handle, &s_) &&
        try_emulate(dtype, (hipblasHandle_t)handle, (hipblasOperation_t)transA, (hipblasOperation_t)transB, m, n, k, alpha, a, lda, b, ldb, beta,
                    d, ldd, &st_, &s_))
        return rocblas_status_of(st_);
    NativeScope ns_; return real ? real(handle, transA, transB, m, n, k, alpha, a, a_type, lda, b, b_type, ldb, beta, c, c_type, ldc, d, d_type, ldd, compute_type,
                       algo, solution_index, flags)
                : 6;
}
}  // extern "C"

// ---- hipblasLtMatmul (not hooked by the reference; PyTorch on ROCm routes most float32 matmuls through hipBLASLt, so without
// this GEMMUL8_NUM_MOD_S is a no-op for them).  Only the plain case is emulated: D = alpha*op(A)*op(B) + beta*C with A, B, C, D of
// one type in {float, double, complex float, complex double}, column-major order, single or strided-batched, default epilogue, no scale pointers,
// host or device scalars; everything else goes to the real routine untouched.  C != D is served in place on D after a copy of C.
#include <hipblaslt/hipblaslt.h>
namespace {
struct LtLayout {
    int32_t type = -1, order = -1, batch = 1;
    uint64_t rows = 0, cols = 0;
    int64_t ld = 0;
    int64_t stride = 0;  // STRIDED_BATCH_OFFSET, elements
};
// hipBLASLt (ROCm 7.2) cannot be asked what a matrix layout holds -- hipblasLtMatrixLayoutGetAttribute answers only the two batch
// attributes -- so the hook records type / rows / cols / ld / order / batch when a layout is created or modified
// (hipblasLtMatrixLayoutCreate / SetAttribute / Destroy are interposed below) and reads its own table here.
std::mutex g_lt_mtx;
std::unordered_map<hipblasLtMatrixLayout_t, LtLayout> g_lt_layouts;
bool lt_layout(hipblasLtMatrixLayout_t L, LtLayout* o) {
    std::lock_guard<std::mutex> g(g_lt_mtx);
    auto it = g_lt_layouts.find(L);
    if (it == g_lt_layouts.end()) return false;  // created before the hook was loaded, or by an interface we do not see
    *o = it->second;
    return true;
}
// GEMMUL8_HOOK_VERBOSE=1: say why a hipblasLtMatmul call was left to the native routine
bool lt_decline(const char* why) {
    if (env_one("GEMMUL8_HOOK_VERBOSE")) std::fprintf(stderr, "[GEMMUL8 HOOK] hipblasLtMatmul -> native: %s\n", why);
    return false;
}
// returns true when the call was emulated (status in *st)
bool lt_try(hipblasLtHandle_t handle, hipblasLtMatmulDesc_t desc, const void* alpha, const void* A, hipblasLtMatrixLayout_t Ad, const void* B,
            hipblasLtMatrixLayout_t Bd, const void* beta, const void* C, hipblasLtMatrixLayout_t Cd, void* D, hipblasLtMatrixLayout_t Dd,
            hipStream_t stream, hipblasStatus_t* st) {
    if (!desc || !alpha || !beta || !A || !B || !D) return lt_decline("null argument");
    using DescGet = hipblasStatus_t (*)(hipblasLtMatmulDesc_t, hipblasLtMatmulDescAttributes_t, void*, size_t, size_t*);
    static DescGet dget = real_fn<DescGet>("hipblasLtMatmulDescGetAttribute");
    if (!dget) return lt_decline("hipblasLtMatmulDescGetAttribute not found");
    size_t w = 0;
    int32_t ta = 0, tb = 0;
    uint32_t epi = 0;
    if (dget(desc, HIPBLASLT_MATMUL_DESC_TRANSA, &ta, sizeof ta, &w) != HIPBLAS_STATUS_SUCCESS) return lt_decline("TRANSA unreadable");
    if (dget(desc, HIPBLASLT_MATMUL_DESC_TRANSB, &tb, sizeof tb, &w) != HIPBLAS_STATUS_SUCCESS) return lt_decline("TRANSB unreadable");
    if (dget(desc, HIPBLASLT_MATMUL_DESC_EPILOGUE, &epi, sizeof epi, &w) != HIPBLAS_STATUS_SUCCESS ||
        (epi != HIPBLASLT_EPILOGUE_DEFAULT && epi != HIPBLASLT_EPILOGUE_BIAS))
        return lt_decline("epilogue is neither the default one nor a plain bias");
    // BIAS epilogue (what a float32 torch.nn.Linear issues): D = alpha*op(A)*op(B) + beta*C + bias, bias broadcast over the columns --
    // emulated as the plain GEMM followed by the bias addition when the bias vector has the matrices' (real) type
    const void* bias = nullptr;
    if (epi == HIPBLASLT_EPILOGUE_BIAS) {
        if (dget(desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof bias, &w) != HIPBLAS_STATUS_SUCCESS || !bias) return lt_decline("bias epilogue without a bias pointer");
    }
    for (auto attr : {HIPBLASLT_MATMUL_DESC_A_SCALE_POINTER, HIPBLASLT_MATMUL_DESC_B_SCALE_POINTER, HIPBLASLT_MATMUL_DESC_C_SCALE_POINTER,
                      HIPBLASLT_MATMUL_DESC_D_SCALE_POINTER, HIPBLASLT_MATMUL_DESC_AMAX_D_POINTER}) {
        void* ptr = nullptr;
        if (dget(desc, attr, &ptr, sizeof ptr, &w) == HIPBLAS_STATUS_SUCCESS && ptr) return lt_decline("scale / amax pointer set");
    }
    int32_t pmode = 0;
    if (dget(desc, HIPBLASLT_MATMUL_DESC_POINTER_MODE, &pmode, sizeof pmode, &w) == HIPBLAS_STATUS_SUCCESS && pmode != HIPBLASLT_POINTER_MODE_HOST &&
        pmode != HIPBLASLT_POINTER_MODE_DEVICE)
        return lt_decline("device-vector scalars");
    LtLayout a, b, c, d;
    if (!lt_layout(Ad, &a) || !lt_layout(Bd, &b) || !lt_layout(Dd, &d)) return lt_decline("a matrix layout was not created under the hook");
    const bool haveC = C && Cd;
    if (haveC && !lt_layout(Cd, &c)) return lt_decline("the C layout was not created under the hook");
    if (!haveC) c = d;
    if (a.type != b.type || a.type != d.type || c.type != d.type) return lt_decline("mixed matrix types");
    int dtype = -1;
    switch (a.type) {
    case HIP_R_32F: dtype = GEMMUL8_S; break;
    case HIP_R_64F: dtype = GEMMUL8_D; break;
    case HIP_C_32F: dtype = GEMMUL8_C; break;
    case HIP_C_64F: dtype = GEMMUL8_Z; break;
    default: return lt_decline("not an S/D/C/Z matrix type");
    }
    if (bias) {
        int32_t bt = a.type;
        // an unset BIAS_DATA_TYPE reads back as 255 (invalid) in ROCm 7.2 and means "the type of D"
        if (dget(desc, HIPBLASLT_MATMUL_DESC_BIAS_DATA_TYPE, &bt, sizeof bt, &w) != HIPBLAS_STATUS_SUCCESS || (bt != a.type && bt != 255))
            return lt_decline("bias vector of another type than the matrices");
        if (dtype != GEMMUL8_S && dtype != GEMMUL8_D) return lt_decline("bias epilogue on a complex type");
        if (d.batch != 1 || d.cols > 65535) return lt_decline("bias epilogue on a batched or very wide product");
    }
    if (a.order != HIPBLASLT_ORDER_COL || b.order != HIPBLASLT_ORDER_COL || c.order != HIPBLASLT_ORDER_COL || d.order != HIPBLASLT_ORDER_COL) return lt_decline("not column-major");
    const int nb = d.batch;
    if (nb < 1 || a.batch != nb || b.batch != nb || c.batch != nb) return lt_decline("batch counts differ between the layouts");
    if (nb > 1 && d.stride == 0) return lt_decline("batched with overlapping outputs");
    const uint64_t m = d.rows, n = d.cols, k = (ta == HIPBLAS_OP_N) ? a.cols : a.rows;
    if ((ta == HIPBLAS_OP_N ? a.rows : a.cols) != m || (tb == HIPBLAS_OP_N ? b.cols : b.rows) != n || (tb == HIPBLAS_OP_N ? b.rows : b.cols) != k)
        return lt_decline("inconsistent dimensions");
    if (c.rows != m || c.cols != n || m == 0 || n == 0 || k == 0) return lt_decline("C / D shape");
    if (!fits_int((int64_t)m, (int64_t)n, (int64_t)k, a.ld, b.ld, d.ld) || c.ld > 2147483647) return false;
    const size_t esz = dtype == GEMMUL8_S ? 4 : dtype == GEMMUL8_Z ? 16 : 8;
    // cheap env test before touching D: is emulation selected for this type at all?
    const unsigned N = (unsigned)env_u64(kTypes[dtype].nmod, 0);
    if (N < 2u || N > kTypes[dtype].max_moduli) return false;
    if (below_floor(dtype, (double)m, (double)n, (double)k, N, env_one(kTypes[dtype].fast), env_backend("GEMMUL8_BACKEND", 0, false), (double)nb)) return false;
    if (k > (1u << 17) || (env_backend("GEMMUL8_BACKEND", 0, false) == 1 && k > 65536)) return false;  // outside the emulator's range
    // C is not read when the host scalar beta is 0 (the CRT's "C = +-AB" forms and its general form with beta == 0, oz2_crt.hip): then
    // the out-of-place form needs no copy of C into D.  Device scalars: beta is unknown here, C is copied (the kernel still skips
    // reading it when *beta == 0).
    bool c_unread = false;
    if (pmode == HIPBLASLT_POINTER_MODE_HOST) {
        double ar, ai = 0, br, bi2 = 0;
        if (dtype == GEMMUL8_S || dtype == GEMMUL8_C) {
            ar = ((const float*)alpha)[0], br = ((const float*)beta)[0];
            if (dtype == GEMMUL8_C) ai = ((const float*)alpha)[1], bi2 = ((const float*)beta)[1];
        } else {
            ar = ((const double*)alpha)[0], br = ((const double*)beta)[0];
            if (dtype == GEMMUL8_Z) ai = ((const double*)alpha)[1], bi2 = ((const double*)beta)[1];
        }
        (void)ar, (void)ai;
        c_unread = br == 0 && bi2 == 0;
    }
    if (haveC && C != D && !c_unread) {  // out-of-place form: bring C into D, then update D in place
        for (int bi = 0; bi < nb; ++bi)
            if (hipMemcpy2DAsync((char*)D + (long long)bi * d.stride * (long long)esz, (size_t)d.ld * esz,
                                 (const char*)C + (long long)bi * c.stride * (long long)esz, (size_t)c.ld * esz, m * esz, n,
                                 hipMemcpyDeviceToDevice, stream) != hipSuccess)
                return *st = HIPBLAS_STATUS_INTERNAL_ERROR, true;
    }
    if (nb > 1)  // strided batch (torch.bmm in float32 arrives here): one set of launches, as for hipblas*gemmStridedBatched
        return emulate_batch(dtype, esz, (hipblasHandle_t)handle, (hipblasOperation_t)ta, (hipblasOperation_t)tb, (int)m, (int)n, (int)k, alpha, A,
                             (int)a.ld, (long long)a.stride, B, (int)b.ld, (long long)b.stride, beta, D, (int)d.ld, (long long)d.stride, nb, st, &stream);
    const bool done = try_emulate(dtype, (hipblasHandle_t)handle, (hipblasOperation_t)ta, (hipblasOperation_t)tb, (int)m, (int)n, (int)k, alpha, A,
                                  (int)a.ld, B, (int)b.ld, beta, D, (int)d.ld, st, &stream);
    if (done && bias && *st == HIPBLAS_STATUS_SUCCESS &&
        gemmul8_add_row_bias(stream, dtype, (size_t)m, (size_t)n, D, (size_t)d.ld, bias) != 0)
        *st = HIPBLAS_STATUS_INTERNAL_ERROR;
    return done;
}
}  // namespace

extern "C" hipblasStatus_t hipblasLtMatrixLayoutCreate(hipblasLtMatrixLayout_t* matLayout, hipDataType type, uint64_t rows, uint64_t cols, int64_t ld) {
    using Fn = hipblasStatus_t (*)(hipblasLtMatrixLayout_t*, hipDataType, uint64_t, uint64_t, int64_t);
    static Fn real = real_fn<Fn>("hipblasLtMatrixLayoutCreate");
    if (!real) return HIPBLAS_STATUS_NOT_INITIALIZED;
    const hipblasStatus_t st = real(matLayout, type, rows, cols, ld);
    if (st == HIPBLAS_STATUS_SUCCESS && matLayout && *matLayout) {
        LtLayout rec;
        rec.type = (int32_t)type, rec.order = HIPBLASLT_ORDER_COL, rec.batch = 1, rec.rows = rows, rec.cols = cols, rec.ld = ld;
        std::lock_guard<std::mutex> g(g_lt_mtx);
        g_lt_layouts[*matLayout] = rec;
    }
    return st;
}
extern "C" hipblasStatus_t hipblasLtMatrixLayoutSetAttribute(hipblasLtMatrixLayout_t matLayout, hipblasLtMatrixLayoutAttribute_t attr, const void* buf,
                                                             size_t sizeInBytes) {
    using Fn = hipblasStatus_t (*)(hipblasLtMatrixLayout_t, hipblasLtMatrixLayoutAttribute_t, const void*, size_t);
    static Fn real = real_fn<Fn>("hipblasLtMatrixLayoutSetAttribute");
    if (!real) return HIPBLAS_STATUS_NOT_INITIALIZED;
    const hipblasStatus_t st = real(matLayout, attr, buf, sizeInBytes);
    if (st == HIPBLAS_STATUS_SUCCESS && buf) {
        std::lock_guard<std::mutex> g(g_lt_mtx);
        auto it = g_lt_layouts.find(matLayout);
        if (it != g_lt_layouts.end()) {
            LtLayout& r = it->second;
            switch (attr) {
            case HIPBLASLT_MATRIX_LAYOUT_BATCH_COUNT: if (sizeInBytes >= 4) std::memcpy(&r.batch, buf, 4); break;
            case HIPBLASLT_MATRIX_LAYOUT_STRIDED_BATCH_OFFSET: if (sizeInBytes >= 8) std::memcpy(&r.stride, buf, 8); break;
            case HIPBLASLT_MATRIX_LAYOUT_TYPE: if (sizeInBytes >= 4) std::memcpy(&r.type, buf, 4); break;
            case HIPBLASLT_MATRIX_LAYOUT_ORDER: if (sizeInBytes >= 4) std::memcpy(&r.order, buf, 4); break;
            case HIPBLASLT_MATRIX_LAYOUT_ROWS: if (sizeInBytes >= 8) std::memcpy(&r.rows, buf, 8); break;
            case HIPBLASLT_MATRIX_LAYOUT_COLS: if (sizeInBytes >= 8) std::memcpy(&r.cols, buf, 8); break;
            case HIPBLASLT_MATRIX_LAYOUT_LD: if (sizeInBytes >= 8) std::memcpy(&r.ld, buf, 8); break;
            default: break;
            }
        }
    }
    return st;
}
extern "C" hipblasStatus_t hipblasLtMatrixLayoutDestroy(const hipblasLtMatrixLayout_t matLayout) {
    {
        std::lock_guard<std::mutex> g(g_lt_mtx);
        g_lt_layouts.erase(matLayout);
    }
    using Fn = hipblasStatus_t (*)(const hipblasLtMatrixLayout_t);
    static Fn real = real_fn<Fn>("hipblasLtMatrixLayoutDestroy");
    NativeScope ns_; return real ? real(matLayout) : HIPBLAS_STATUS_NOT_INITIALIZED;
}

// the per-handle state of an hipblasLt handle is released with the handle, as for hipblasDestroy
extern "C" hipblasStatus_t hipblasLtDestroy(const hipblasLtHandle_t handle) {
    release_state((hipblasHandle_t)handle, true);
    using Fn = hipblasStatus_t (*)(const hipblasLtHandle_t);
    static Fn real = real_fn<Fn>("hipblasLtDestroy");
    NativeScope ns_; return real ? real(handle) : HIPBLAS_STATUS_NOT_INITIALIZED;
}

extern "C" hipblasStatus_t hipblasLtMatmul(hipblasLtHandle_t handle, hipblasLtMatmulDesc_t matmulDesc, const void* alpha, const void* A,
                                           hipblasLtMatrixLayout_t Adesc, const void* B, hipblasLtMatrixLayout_t Bdesc, const void* beta,
                                           const void* C, hipblasLtMatrixLayout_t Cdesc, void* D, hipblasLtMatrixLayout_t Ddesc,
                                           const hipblasLtMatmulAlgo_t* algo, void* workspace, size_t workspaceSizeInBytes, hipStream_t stream) {
    hipblasStatus_t st;
    if (lt_try(handle, matmulDesc, alpha, A, Adesc, B, Bdesc, beta, C, Cdesc, D, Ddesc, stream, &st)) return st;
    using Fn = hipblasStatus_t (*)(hipblasLtHandle_t, hipblasLtMatmulDesc_t, const void*, const void*, hipblasLtMatrixLayout_t, const void*,
                                   hipblasLtMatrixLayout_t, const void*, const void*, hipblasLtMatrixLayout_t, void*, hipblasLtMatrixLayout_t,
                                   const hipblasLtMatmulAlgo_t*, void*, size_t, hipStream_t);
    static Fn real = real_fn<Fn>("hipblasLtMatmul");
    NativeScope ns_; return real ? real(handle, matmulDesc, alpha, A, Adesc, B, Bdesc, beta, C, Cdesc, D, Ddesc, algo, workspace, workspaceSizeInBytes, stream)
                : HIPBLAS_STATUS_NOT_INITIALIZED;
}
#pragma GCC visibility pop
