/*
 * gemmul8_dist.h -- C ABI of the multi-GPU (one process per GPU) emulated GEMM of libgemmul8.so.
 *
 * The reference has no multi-GPU code; BASELINE.json's north_star asks for the moduli pipelines to be
 * sharded over the GPUs of one node with an RCCL exchange over xGMI, host code in C++ behind a C ABI.  This header is that
 * boundary: a C++ caller of gemmul8::gemm (include/gemmul8.hpp) or an application under the LD_PRELOAD hook
 * (GEMMUL8_DIST=...) reaches the sharded path through it; gemmul8_amd/dist.py is a ctypes caller of it.
 *
 * Placement contract: A and B are REPLICATED (every rank holds the full operands, same values -- e.g. same seed, or broadcast
 * by the caller), C is a full-size matrix on every rank of which a rank updates only the part it owns
 * (gemmul8_dist_owned_block); gemmul8_dist_allgather_c assembles the whole result on every rank when a caller needs it.
 *
 * Three plans, selected at plan creation:
 *   GEMMUL8_DIST_BLOCKS          rank (i, j) of a Gr x Gc grid owns C[rows_i, cols_j] and runs ALL moduli on that block through the
 *                                single-GPU phase entry points (gemmul8_c.h) on strided views of A and B.  Only coupling: the
 *                                accurate mode's bound maxima -- ONE all-reduce(MAX) of int32[m + n].  Bit-identical to one GPU.
 *   GEMMUL8_DIST_MODULI          rank r owns moduli [t0_r, t1_r) and columns [c0_r, c1_r): bound GEMM on its columns +
 *                                all-reduce(MAX), quantise + low-precision GEMMs for its moduli, point-to-point exchange of INT8
 *                                residue blocks over xGMI, reference-order CRT on its columns.  Bit-identical to one GPU.
 *   GEMMUL8_DIST_MODULI_FP64SUM  the exchange north_star names: each rank forms the FP64 (double-double) partial CRT sum of its
 *                                moduli, the ranks reduce-scatter(sum) the partials, the owner finishes.  Integer
 *                                intermediates identical; the final FP value can differ from one GPU in the last bits (the
 *                                rounded lo chain is grouped by rank) -- measured in tests/test_gpu_dist.py and DESIGN.md 5.
 */
#ifndef GEMMUL8_DIST_H
#define GEMMUL8_DIST_H

#include "gemmul8_c.h"

#ifdef __cplusplus
extern "C" {
#endif

enum { GEMMUL8_DIST_BLOCKS = 0, GEMMUL8_DIST_MODULI = 1, GEMMUL8_DIST_MODULI_FP64SUM = 2 };

/* One message of a grouped point-to-point exchange (device memory). */
typedef struct gemmul8_p2p_op {
    void *buf;
    size_t bytes;
    int peer;    /* rank inside the communicator */
    int is_send; /* 1 = send, 0 = receive */
} gemmul8_p2p_op;

/* Transport: a table of collectives on DEVICE buffers, enqueued on (or completed before returning with respect to) `stream`.
 * The product transport is RCCL (gemmul8_comm_rccl_*); a host may supply any other implementation -- the CPU tests plug
 * torch.distributed/gloo in through ctypes callbacks.  Every function returns 0 on success. */
typedef struct gemmul8_comm {
    void *ctx;
    int rank, world;
    int (*allreduce_max_i32)(void *ctx, void *buf, size_t count, void *stream);                       /* in place */
    int (*sendrecv)(void *ctx, int nops, const gemmul8_p2p_op *ops, void *stream);                    /* one group; no self-sends */
    int (*reduce_scatter_sum_f64)(void *ctx, const void *send, void *recv, size_t recv_count, void *stream); /* send: world * recv_count doubles */
    void (*destroy)(void *ctx);
} gemmul8_comm;

/* RCCL transport.  librccl is opened at run time (the copy already mapped into the process if there is one -- PyTorch bundles
 * its own next to its HIP runtime -- else librccl.so.1); libgemmul8.so has no link-time dependency on it.
 *   _unique_id : rank 0 fills 128 bytes (ncclUniqueId) that the caller distributes to the other ranks by any means;
 *   _create    : collective over all ranks (ncclCommInitRank on the current device);
 *   _from_env  : both steps for a launcher that sets RANK, WORLD_SIZE, MASTER_ADDR and MASTER_PORT (torchrun, mpirun wrappers):
 *                the id travels over a TCP connection to MASTER_ADDR:GEMMUL8_DIST_PORT (default MASTER_PORT + 17). */
GEMMUL8_API int gemmul8_comm_rccl_unique_id(void *id128);
GEMMUL8_API int gemmul8_comm_rccl_create(const void *id128, int rank, int world, gemmul8_comm **out);
GEMMUL8_API int gemmul8_comm_rccl_id_from_env(void *id128, int *rank, int *world); /* the rendezvous alone: same 128 bytes on every rank */
GEMMUL8_API int gemmul8_comm_rccl_from_env(gemmul8_comm **out);
GEMMUL8_API void gemmul8_comm_destroy(gemmul8_comm *comm);
/* Size of the RCCL communicator as the library reports it (ncclCommCount): *count = -1 when `comm` is not an RCCL transport. */
GEMMUL8_API int gemmul8_comm_rccl_count(const gemmul8_comm *comm, int *count);

/* Compute + memory provider of a plan.  NULL (the product) = the HIP phase entry points of gemmul8_c.h and hipMalloc /
 * hipMemcpyAsync / hipMemsetAsync.  The table exists so that the sharding and exchange arithmetic of this file -- which never
 * sees more than one GPU in this build environment -- can be driven at world sizes 2..8 on CPU by the test-suite (host memory,
 * the CPU oracle as the engine); signatures are those of gemmul8_c.h. */
typedef struct gemmul8_dist_engine {
    void *(*alloc)(size_t bytes);
    void (*release)(void *p);
    int (*zero)(void *p, size_t bytes, void *stream);
    int (*copy)(void *dst, const void *src, size_t bytes, void *stream);
    int (*copy2d)(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width_bytes, size_t height, void *stream);
    int (*scale_bounds)(void *stream, int dtype, int backend, int op_A, int op_B, size_t m, size_t n, size_t k, const void *A, size_t lda,
                        const void *B, size_t ldb, unsigned num_moduli, size_t col_begin, size_t col_end, const gemmul8_layout *L,
                        int skipA, int skipB);
    int (*scale_finish)(void *stream, int dtype, int backend, int op_A, int op_B, size_t m, size_t n, size_t k, const void *A, size_t lda,
                        const void *B, size_t ldb, unsigned num_moduli, int fastmode, unsigned t_begin, unsigned t_end,
                        const gemmul8_layout *L, int skipA, int skipB);
    int (*lowprec_gemm)(void *stream, int dtype, int backend, size_t m, size_t n, size_t k, unsigned num_moduli, unsigned t_begin,
                        unsigned t_end, const gemmul8_layout *L);
    int (*crt)(void *stream, int dtype, int backend, unsigned num_moduli, size_t m, size_t n, const void *C_mid, size_t ld_mid,
               size_t plane_stride, const int16_t *sftA, const int16_t *sftB, const void *alpha, const void *beta, void *C, size_t ldc);
    int (*crt_partial)(void *stream, int dtype, int backend, unsigned num_moduli, unsigned t_begin, unsigned t_end, size_t m, size_t n,
                       const void *C_mid, size_t ld_mid, size_t plane_stride, double *out_hi, double *out_lo, size_t ld_out,
                       size_t col_block, size_t block_stride);
    int (*crt_finish)(void *stream, int dtype, int backend, unsigned num_moduli, size_t m, size_t n, const double *in_hi,
                      const double *in_lo, size_t ld_in, const int16_t *sftA, const int16_t *sftB, const void *alpha, const void *beta,
                      void *C, size_t ldc);
    /* ABI version 7: dst += src on doubles (gemmul8_add_f64).  May be NULL: the fp64sum plan then exchanges its partial sums in one
     * collective whatever GEMMUL8_DIST_FP64_GROUPS says. */
    int (*add_f64)(void *stream, double *dst, const double *src, size_t count);
} gemmul8_dist_engine;

typedef struct gemmul8_dist_plan gemmul8_dist_plan;

/* Plan for C = alpha*op(A)*op(B) + beta*C of fixed shape on the ranks of `comm` (not owned by the plan).  grid_rows = 0 picks
 * the rank grid of the block plan (Gr >= Gc as square as possible: 2 -> 2x1, 4 -> 2x2, 8 -> 4x2); otherwise Gr = grid_rows must
 * divide the world size.  Allocates the plan's workspaces through the engine.  Collective in the sense that every rank must
 * create the same plan; performs no communication. */
GEMMUL8_API int gemmul8_dist_create(const gemmul8_comm *comm, const gemmul8_dist_engine *engine, int plan, int grid_rows, int dtype,
                                    int backend, int op_A, int op_B, size_t m, size_t n, size_t k, unsigned num_moduli, int fastmode,
                                    gemmul8_dist_plan **out);
/* One sharded GEMM; asynchronous on `stream` with the RCCL transport.  alpha / beta: host or device pointers. */
GEMMUL8_API int gemmul8_dist_gemm(gemmul8_dist_plan *plan, void *stream, const void *alpha, const void *A, size_t lda, const void *B,
                                  size_t ldb, const void *beta, void *C, size_t ldc);
/* The part of C this rank updates: rows [r0, r1) x columns [c0, c1) (empty ranges possible when m or n < the grid). */
GEMMUL8_API int gemmul8_dist_owned_block(const gemmul8_dist_plan *plan, int rank, size_t *r0, size_t *r1, size_t *c0, size_t *c1);
/* Number of low-precision planes (moduli) this rank multiplies and the 2*m*n*k-units of its share (for roofline accounting). */
GEMMUL8_API int gemmul8_dist_my_work(const gemmul8_dist_plan *plan, unsigned *moduli, size_t *rows, size_t *cols);
/* Assemble the full C on every rank from the owned blocks (verification, or callers with replicated semantics such as the hook). */
GEMMUL8_API int gemmul8_dist_allgather_c(gemmul8_dist_plan *plan, void *stream, void *C, size_t ldc);
/* Measurement hook: two hipEvent_t (or NULL) that the next gemmul8_dist_gemm calls record on their stream right before and
 * after the low-precision GEMM launch of this rank (the dominant kernel: bench.py's roofline line). */
GEMMUL8_API int gemmul8_dist_set_events(gemmul8_dist_plan *plan, void *ev_begin, void *ev_end);
/* The same for the plan's collectives: ev[0], ev[1] around the all-reduce(MAX) of the bound maxima (accurate mode, world > 1),
 * ev[2], ev[3] around the bulk exchange (moduli: grouped send/recv of residue blocks; fp64sum: reduce-scatter; blocks: none).
 * ev = NULL clears; entries may be NULL.  An event that the call does not reach is left untouched.
 * Moduli plan with GEMMUL8_DIST_GROUPS > 1: the exchange runs on the plan's own stream BESIDE the GEMMs of the later groups, so
 * ev[2] .. ev[3] (first group's exchange begins .. last group's ends) spans GEMM time as well and must not be added to the GEMM phase;
 * the EXPOSED exchange time is elapsed(ev_end of gemmul8_dist_set_events, ev[3]).  The first call of such a plan synchronises `stream`
 * once to cross-check GEMMUL8_DIST_GROUPS between the ranks (a mismatch is GEMMUL8_E_ARG on every rank, not a hang). */
GEMMUL8_API int gemmul8_dist_set_exchange_events(gemmul8_dist_plan *plan, void *const ev[4]);
/* Bytes this rank moves per gemmul8_dist_gemm call: the all-reduce payload (bytes of the reduced vector), and what the bulk exchange
 * sends to / receives from other ranks (fp64sum: the (world - 1) / world share of the reduce-scatter input / its output). */
GEMMUL8_API int gemmul8_dist_exchange_bytes(const gemmul8_dist_plan *plan, size_t *allreduce_bytes, size_t *sent, size_t *received);
GEMMUL8_API size_t gemmul8_dist_workspace_bytes(const gemmul8_dist_plan *plan);
GEMMUL8_API void gemmul8_dist_destroy(gemmul8_dist_plan *plan);

#ifdef __cplusplus
}
#endif
#endif
