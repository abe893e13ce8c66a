// Drives every entry point of the LD_PRELOAD hook (gemmul8_amd/csrc/oz2_hook.cpp, built with ASan + UBSan) against the mock
// libraries of mock_gpu.cpp: growing shapes, stream switches, the skip-scaling cache, type switches, the ILP64 / Ex / batched /
// hipblasLt forms, calls outside the emulator's range, the size floor, hipblasDestroy, and several host threads with their own handles.
#include <hip/hip_runtime.h>
#include <hipblas/hipblas.h>
#include <hipblaslt/hipblaslt.h>

#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

extern "C" {
long mock_native_calls();
long mock_emulated_calls();
long mock_live_allocs();
hipblasStatus_t mock_create(hipblasHandle_t*);
hipblasStatus_t mock_set_stream(hipblasHandle_t, hipStream_t);
void* mock_lt_desc(int, int, unsigned);
void* mock_lt_desc_bias(int, int, const void*, int);
void mock_lt_free_desc(void*);
}
#define CHECK(x)                                                       \
    do {                                                               \
        if (!(x)) {                                                    \
            std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #x); \
            std::exit(1);                                              \
        }                                                              \
    } while (0)

static void one_thread(int id) {
    hipblasHandle_t h;
    mock_create(&h);
    const double one = 1, zero = 0;
    const float onef = 1, zerof = 0;
    std::vector<double> A(600 * 600), B(600 * 600), C(600 * 600);
    std::vector<float> fA(300 * 300), fB(300 * 300), fC(300 * 300);
    const int shapes[][3] = {{64, 48, 100}, {300, 200, 520}, {64, 48, 100}, {600, 600, 257}, {1, 1, 1}, {300, 200, 520}};
    for (int rep = 0; rep < 3; ++rep)
        for (auto& s : shapes) {
            if (rep == 1) mock_set_stream(h, (hipStream_t)(uintptr_t)(0x1000 + id));  // stream switch -> event hand-off
            CHECK(hipblasDgemm(h, HIPBLAS_OP_N, HIPBLAS_OP_T, s[0], s[1], s[2], &one, A.data(), s[0], B.data(), s[1], &zero, C.data(), s[0]) == HIPBLAS_STATUS_SUCCESS);
            CHECK(hipblasDgemm(h, HIPBLAS_OP_N, HIPBLAS_OP_T, s[0], s[1], s[2], &one, A.data(), s[0], B.data(), s[1], &zero, C.data(), s[0]) == HIPBLAS_STATUS_SUCCESS);  // same pointers: skip cache
            CHECK(hipblasSgemm(h, HIPBLAS_OP_T, HIPBLAS_OP_N, 200, 150, 300, &onef, fA.data(), 300, fB.data(), 300, &zerof, fC.data(), 200) == HIPBLAS_STATUS_SUCCESS);
        }
    CHECK(hipblasDgemm_64(h, HIPBLAS_OP_N, HIPBLAS_OP_N, 100, 90, 80, &one, A.data(), 100, B.data(), 80, &zero, C.data(), 100) == HIPBLAS_STATUS_SUCCESS);
    CHECK(hipblasGemmEx(h, HIPBLAS_OP_N, HIPBLAS_OP_N, 100, 90, 80, &one, A.data(), HIP_R_64F, 100, B.data(), HIP_R_64F, 80, &zero, C.data(), HIP_R_64F, 100,
                        HIPBLAS_COMPUTE_64F, HIPBLAS_GEMM_DEFAULT) == HIPBLAS_STATUS_SUCCESS);
    CHECK(hipblasGemmEx(h, HIPBLAS_OP_N, HIPBLAS_OP_N, 100, 90, 80, &one, A.data(), HIP_R_16F, 100, B.data(), HIP_R_16F, 80, &zero, C.data(), HIP_R_16F, 100,
                        HIPBLAS_COMPUTE_32F, HIPBLAS_GEMM_DEFAULT) == HIPBLAS_STATUS_SUCCESS);  // not an emulated type: native
    {  // batched: one set of launches (INT8 backend default), then 7 items over the default 4 lanes (own streams and workspaces,
       // fork / join events), then serial with one lane
        const long emu_a = mock_emulated_calls();
        CHECK(hipblasDgemmStridedBatched(h, HIPBLAS_OP_N, HIPBLAS_OP_N, 50, 40, 30, &one, A.data(), 50, 1500, B.data(), 30, 1200, &zero, C.data(), 50, 2000, 7) ==
              HIPBLAS_STATUS_SUCCESS);
        CHECK(mock_emulated_calls() >= emu_a + 7);
        // the environment is process state and setenv is not thread-safe against getenv: only the first, sequential pass (id 0)
        // switches the batch paths; the concurrent passes run the default one
        if (id != 0) goto batch_done;
        setenv("GEMMUL8_BATCH_WORKSPACE_MB", "1", 1);  // smaller than one item: chunks of a single item, the buffer is re-grown
        const long emu_c = mock_emulated_calls();
        CHECK(hipblasDgemmStridedBatched(h, HIPBLAS_OP_N, HIPBLAS_OP_N, 50, 40, 30, &one, A.data(), 50, 1500, B.data(), 30, 1200, &zero, C.data(), 50, 2000, 5) ==
              HIPBLAS_STATUS_SUCCESS);
        CHECK(mock_emulated_calls() >= emu_c + 5);
        unsetenv("GEMMUL8_BATCH_WORKSPACE_MB");
        setenv("GEMMUL8_BATCH_FUSED", "0", 1);
        const long emu_b = mock_emulated_calls();
        CHECK(hipblasDgemmStridedBatched(h, HIPBLAS_OP_N, HIPBLAS_OP_N, 50, 40, 30, &one, A.data(), 50, 1500, B.data(), 30, 1200, &zero, C.data(), 50, 2000, 7) ==
              HIPBLAS_STATUS_SUCCESS);
        CHECK(mock_emulated_calls() >= emu_b + 7);  // (other threads count too)
        setenv("GEMMUL8_BATCH_STREAMS", "1", 1);
        CHECK(hipblasDgemmStridedBatched(h, HIPBLAS_OP_N, HIPBLAS_OP_N, 50, 40, 30, &one, A.data(), 50, 1500, B.data(), 30, 1200, &zero, C.data(), 50, 2000, 3) ==
              HIPBLAS_STATUS_SUCCESS);
        unsetenv("GEMMUL8_BATCH_STREAMS");
        unsetenv("GEMMUL8_BATCH_FUSED");
    }
batch_done:
    // outside the emulator's range (k > 2^17): native, not an error
    std::vector<double> Ak((size_t)4 * ((1 << 17) + 8)), Bk((size_t)((1 << 17) + 8) * 3);
    const long nat0 = mock_native_calls();
    CHECK(hipblasDgemm(h, HIPBLAS_OP_N, HIPBLAS_OP_N, 4, 3, (1 << 17) + 8, &one, Ak.data(), 4, Bk.data(), (1 << 17) + 8, &zero, C.data(), 4) == HIPBLAS_STATUS_SUCCESS);
    CHECK(mock_native_calls() > nat0);
    // early outs
    CHECK(hipblasDgemm(h, HIPBLAS_OP_N, HIPBLAS_OP_N, 0, 3, 5, &one, A.data(), 1, B.data(), 5, &zero, C.data(), 1) == HIPBLAS_STATUS_SUCCESS);
    CHECK(hipblasDgemm(h, HIPBLAS_OP_N, HIPBLAS_OP_N, 4, 3, 5, &one, nullptr, 4, B.data(), 5, &zero, C.data(), 4) == HIPBLAS_STATUS_INVALID_VALUE);
    // hipblasLt: plain double matmul out of place (emulated), batch 2 (native), bias epilogue (native)
    hipblasLtHandle_t lt = (hipblasLtHandle_t)(uintptr_t)(0x9000 + id);
    void* d = mock_lt_desc(HIPBLAS_OP_T, HIPBLAS_OP_N, HIPBLASLT_EPILOGUE_DEFAULT);
    hipblasLtMatrixLayout_t la, lb, lb2, lc, ld;  // created through the hooked entry points: the hook records what they hold
    CHECK(hipblasLtMatrixLayoutCreate(&la, HIP_R_64F, 80, 100, 80) == HIPBLAS_STATUS_SUCCESS);
    CHECK(hipblasLtMatrixLayoutCreate(&lb, HIP_R_64F, 80, 90, 80) == HIPBLAS_STATUS_SUCCESS);
    CHECK(hipblasLtMatrixLayoutCreate(&lb2, HIP_R_64F, 80, 90, 80) == HIPBLAS_STATUS_SUCCESS);
    CHECK(hipblasLtMatrixLayoutCreate(&lc, HIP_R_64F, 100, 90, 100) == HIPBLAS_STATUS_SUCCESS);
    CHECK(hipblasLtMatrixLayoutCreate(&ld, HIP_R_64F, 100, 90, 104) == HIPBLAS_STATUS_SUCCESS);
    const int32_t two = 2;
    CHECK(hipblasLtMatrixLayoutSetAttribute(lb2, HIPBLASLT_MATRIX_LAYOUT_BATCH_COUNT, &two, sizeof two) == HIPBLAS_STATUS_SUCCESS);
    std::vector<double> D(104 * 90);
    const long emu0 = mock_emulated_calls();
    CHECK(hipblasLtMatmul(lt, (hipblasLtMatmulDesc_t)d, &one, A.data(), (hipblasLtMatrixLayout_t)la, B.data(), (hipblasLtMatrixLayout_t)lb, &one, C.data(),
                          (hipblasLtMatrixLayout_t)lc, D.data(), (hipblasLtMatrixLayout_t)ld, nullptr, nullptr, 0, (hipStream_t)(uintptr_t)(0x2000 + id)) == HIPBLAS_STATUS_SUCCESS);
    CHECK(mock_emulated_calls() > emu0);
    const long nat1 = mock_native_calls();
    CHECK(hipblasLtMatmul(lt, (hipblasLtMatmulDesc_t)d, &one, A.data(), (hipblasLtMatrixLayout_t)la, B.data(), (hipblasLtMatrixLayout_t)lb2, &one, C.data(),
                          (hipblasLtMatrixLayout_t)lc, D.data(), (hipblasLtMatrixLayout_t)ld, nullptr, nullptr, 0, nullptr) == HIPBLAS_STATUS_SUCCESS);
    {  // strided batch through hipblasLt (all four layouts with 3 items): emulated as one set of launches
        hipblasLtMatrixLayout_t ba, bb, bc, bd;
        CHECK(hipblasLtMatrixLayoutCreate(&ba, HIP_R_64F, 80, 100, 80) == HIPBLAS_STATUS_SUCCESS);
        CHECK(hipblasLtMatrixLayoutCreate(&bb, HIP_R_64F, 80, 90, 80) == HIPBLAS_STATUS_SUCCESS);
        CHECK(hipblasLtMatrixLayoutCreate(&bc, HIP_R_64F, 100, 90, 100) == HIPBLAS_STATUS_SUCCESS);
        CHECK(hipblasLtMatrixLayoutCreate(&bd, HIP_R_64F, 100, 90, 104) == HIPBLAS_STATUS_SUCCESS);
        const int32_t three = 3;
        const int64_t strides[4] = {8000, 7200, 9000, 9360};
        hipblasLtMatrixLayout_t ls[4] = {ba, bb, bc, bd};
        for (int i = 0; i < 4; ++i) {
            CHECK(hipblasLtMatrixLayoutSetAttribute(ls[i], HIPBLASLT_MATRIX_LAYOUT_BATCH_COUNT, &three, sizeof three) == HIPBLAS_STATUS_SUCCESS);
            CHECK(hipblasLtMatrixLayoutSetAttribute(ls[i], HIPBLASLT_MATRIX_LAYOUT_STRIDED_BATCH_OFFSET, &strides[i], sizeof(int64_t)) == HIPBLAS_STATUS_SUCCESS);
        }
        std::vector<double> D3(3 * 9360);
        const long emu1 = mock_emulated_calls();
        CHECK(hipblasLtMatmul(lt, (hipblasLtMatmulDesc_t)d, &one, A.data(), ba, B.data(), bb, &one, C.data(), bc, D3.data(), bd, nullptr, nullptr, 0,
                              (hipStream_t)(uintptr_t)(0x2000 + id)) == HIPBLAS_STATUS_SUCCESS);
        CHECK(mock_emulated_calls() >= emu1 + 3);
        for (hipblasLtMatrixLayout_t l : ls) CHECK(hipblasLtMatrixLayoutDestroy(l) == HIPBLAS_STATUS_SUCCESS);
    }
    {  // bias epilogue with a bias vector of the matrices' type: emulated GEMM + bias addition
        std::vector<double> bias(100, 0.5);
        void* dbb = mock_lt_desc_bias(HIPBLAS_OP_T, HIPBLAS_OP_N, bias.data(), (int)HIP_R_64F);
        const long emu2 = mock_emulated_calls();
        CHECK(hipblasLtMatmul(lt, (hipblasLtMatmulDesc_t)dbb, &one, A.data(), (hipblasLtMatrixLayout_t)la, B.data(), (hipblasLtMatrixLayout_t)lb, &zero, C.data(),
                              (hipblasLtMatrixLayout_t)lc, D.data(), (hipblasLtMatrixLayout_t)ld, nullptr, nullptr, 0, (hipStream_t)(uintptr_t)(0x2000 + id)) == HIPBLAS_STATUS_SUCCESS);
        CHECK(mock_emulated_calls() > emu2);
        CHECK(((const unsigned char*)D.data())[0] == 0x66);  // the mock's bias pass ran last
        mock_lt_free_desc(dbb);
    }
    void* db = mock_lt_desc(HIPBLAS_OP_T, HIPBLAS_OP_N, HIPBLASLT_EPILOGUE_BIAS);  // no bias pointer: left to the native routine
    CHECK(hipblasLtMatmul(lt, (hipblasLtMatmulDesc_t)db, &one, A.data(), (hipblasLtMatrixLayout_t)la, B.data(), (hipblasLtMatrixLayout_t)lb, &one, C.data(),
                          (hipblasLtMatrixLayout_t)lc, D.data(), (hipblasLtMatrixLayout_t)ld, nullptr, nullptr, 0, nullptr) == HIPBLAS_STATUS_SUCCESS);
    CHECK(mock_native_calls() >= nat1 + 2);
    mock_lt_free_desc(d), mock_lt_free_desc(db);
    for (hipblasLtMatrixLayout_t l : {la, lb, lb2, lc, ld}) CHECK(hipblasLtMatrixLayoutDestroy(l) == HIPBLAS_STATUS_SUCCESS);
    CHECK(hipblasLtDestroy(lt) == HIPBLAS_STATUS_SUCCESS);
    CHECK(hipblasDestroy(h) == HIPBLAS_STATUS_SUCCESS);  // frees the handle's three buffers first
}

int main() {
    setenv("GEMMUL8_NUM_MOD_D", "15", 1);
    setenv("GEMMUL8_NUM_MOD_S", "8", 1);
    setenv("GEMMUL8_SKIP_SCALE_A", "1", 1);
    setenv("GEMMUL8_SKIP_SCALE_B", "1", 1);
    setenv("GEMMUL8_MAX_M", "256", 1);
    setenv("GEMMUL8_MAX_N", "256", 1);
    setenv("GEMMUL8_MAX_K", "512", 1);
    setenv("GEMMUL8_MAX_NUM_MOD", "15", 1);
    one_thread(0);
    std::vector<std::thread> ts;
    for (int i = 1; i <= 4; ++i) ts.emplace_back(one_thread, i);
    for (auto& t : ts) t.join();
    // the size floor and the environment switch are read on every call
    hipblasHandle_t h;
    mock_create(&h);
    const double one = 1, zero = 0;
    std::vector<double> A(64 * 64), B(64 * 64), C(64 * 64);
    setenv("GEMMUL8_MIN_FLOPS", "1000000000", 1);
    long nat = mock_native_calls();
    CHECK(hipblasDgemm(h, HIPBLAS_OP_N, HIPBLAS_OP_N, 64, 64, 64, &one, A.data(), 64, B.data(), 64, &zero, C.data(), 64) == HIPBLAS_STATUS_SUCCESS);
    CHECK(mock_native_calls() == nat + 1);
    setenv("GEMMUL8_MIN_FLOPS", "auto", 1);  // the opt-in fitted cost model: a 64^3 call is far below the measured crossover -> native again
    nat = mock_native_calls();
    CHECK(hipblasDgemm(h, HIPBLAS_OP_N, HIPBLAS_OP_N, 64, 64, 64, &one, A.data(), 64, B.data(), 64, &zero, C.data(), 64) == HIPBLAS_STATUS_SUCCESS);
    CHECK(mock_native_calls() == nat + 1);
    unsetenv("GEMMUL8_MIN_FLOPS");  // unset = emulate every selected call (the reference's behaviour)
    {
        const long emu0 = mock_emulated_calls();
        CHECK(hipblasDgemm(h, HIPBLAS_OP_N, HIPBLAS_OP_N, 64, 64, 64, &one, A.data(), 64, B.data(), 64, &zero, C.data(), 64) == HIPBLAS_STATUS_SUCCESS);
        CHECK(mock_emulated_calls() == emu0 + 1);
    }
    setenv("GEMMUL8_DIST", "blocks", 1);  // no RANK / WORLD_SIZE: one warning, then single-GPU emulation
    long emu = mock_emulated_calls();
    CHECK(hipblasDgemm(h, HIPBLAS_OP_N, HIPBLAS_OP_N, 64, 64, 64, &one, A.data(), 64, B.data(), 64, &zero, C.data(), 64) == HIPBLAS_STATUS_SUCCESS);
    CHECK(mock_emulated_calls() == emu + 1);
    CHECK(hipblasDestroy(h) == HIPBLAS_STATUS_SUCCESS);
    CHECK(mock_live_allocs() == 0);  // every workspace of every destroyed handle was released
    std::printf("hook under sanitizers: %ld emulated, %ld native calls ALL OK\n", mock_emulated_calls(), mock_native_calls());
    return 0;
}
