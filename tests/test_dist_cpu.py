"""CPU (gloo, world_size 2 and 3) tests of the moduli-sharded multi-GPU driver gemmul8_amd.dist:
partitioning, the bound all-reduce(MAX), the residue all-to-all and the column-block CRT are
exercised with the CPU oracle plugged in as the compute engine; the assembled result must be
bit-identical to the single-process oracle for every world size."""
import ctypes as C
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib as ol


class OracleEngine:
    """Same phase interface as gemmul8_amd.dist.HipEngine, computed by oracle/ on CPU (INT8, real)."""

    def __init__(self, A, B, N, fastmode):
        self.A, self.B = np.asfortranarray(A), np.asfortranarray(B)
        self.m, self.k = A.shape
        self.n = B.shape[1]
        self.N, self.fast = N, fastmode
        self.dt = ol.DT[self.A.dtype]
        self.mp = self.m  # no padding in the oracle's planes
        self.mid_bytes = 1
        self.lib = ol.lib()
        self.lib.oz2_bound_maxima_i8.argtypes = [C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t,
                                                 C.c_size_t, C.c_void_p, C.c_void_p]
        self.lib.oz2_shift_finalize_i8.argtypes = [C.c_int, C.c_uint, C.c_size_t, C.c_void_p, C.c_void_p]
        self.mx = torch.zeros(self.m + self.n, dtype=torch.int32)
        self.sftA = np.zeros(self.m, np.int16)
        self.sftB = np.zeros(self.n, np.int16)
        self.A_lo = np.zeros((N, self.m, self.k), np.uint8)
        self.B_lo = np.zeros((N, self.n, self.k), np.uint8)
        self.C_mid = np.zeros((N, self.n, self.m), np.int8)

    def maxima(self):
        return self.mx

    def bounds(self, A, B, c0, c1):
        Ab = np.zeros((self.m, self.k), np.uint8)
        Bb = np.zeros((self.n, self.k), np.uint8)
        self.lib.oz2_extract(self.dt, 0, 0, 0, self.m, self.k, ol._p(self.A), self.A.shape[0], ol._p(Ab), ol._p(self.sftA))
        self.lib.oz2_extract(self.dt, 0, 1, 0, self.n, self.k, ol._p(self.B), self.B.shape[0], ol._p(Bb), ol._p(self.sftB))
        self.mx.zero_()
        mxn = self.mx.numpy()
        self.lib.oz2_bound_maxima_i8(0, self.m, self.n, self.k, ol._p(Ab), ol._p(Bb), c0, c1, mxn[:self.m].ctypes.data,
                                     mxn[self.m:].ctypes.data)

    def finish(self, A, B, t0, t1):
        if self.fast:
            self.lib.oz2_fast_shifts(self.dt, 0, self.N, 0, self.m, self.k, ol._p(self.A), self.A.shape[0], ol._p(self.sftA))
            self.lib.oz2_fast_shifts(self.dt, 0, self.N, 1, self.n, self.k, ol._p(self.B), self.B.shape[0], ol._p(self.sftB))
        else:
            mxn = self.mx.numpy()
            self.lib.oz2_shift_finalize_i8(0, self.N, self.m, mxn[:self.m].ctypes.data, ol._p(self.sftA))
            self.lib.oz2_shift_finalize_i8(0, self.N, self.n, mxn[self.m:].ctypes.data, ol._p(self.sftB))
        # the oracle quantises all planes; planes outside [t0,t1) are poisoned to prove they are never used
        self.lib.oz2_quantise(self.dt, 0, self.N, 0, 0, self.m, self.k, ol._p(self.A), self.A.shape[0], ol._p(self.sftA), ol._p(self.A_lo))
        self.lib.oz2_quantise(self.dt, 0, self.N, 1, 0, self.n, self.k, ol._p(self.B), self.B.shape[0], ol._p(self.sftB), ol._p(self.B_lo))
        for t in range(self.N):
            if not (t0 <= t < t1):
                self.A_lo[t] = 0x55
                self.B_lo[t] = 0x33

    def lowprec(self, t0, t1):
        self.C_mid[:] = 77  # poison
        self.lib.oz2_gemm_mod(0, 0, self.N, self.m, self.n, self.k, ol._p(self.A_lo), ol._p(self.B_lo), ol._p(self.C_mid), t0, t1)

    def plane_block(self, t, c0, c1):
        return torch.from_numpy(self.C_mid[t, c0:c1].reshape(-1).view(np.uint8))

    def new_recv(self, ncols):
        return torch.zeros(self.N * ncols * self.m, dtype=torch.uint8)

    def crt(self, recv, c0, c1, Cmat, alpha_ptr, beta_ptr):
        ncols = c1 - c0
        if ncols == 0:
            return
        mid = recv.numpy().view(np.int8)
        Cn = Cmat.numpy()  # (n, m) tensor == column-major m x n
        blk = Cn[c0:c1]
        self.lib.oz2_invscal(self.dt, 0, self.N, self.m, ncols, mid.ctypes.data, ol._p(self.sftA), self.sftB[c0:c1].ctypes.data,
                             alpha_ptr, beta_ptr, blk.ctypes.data, self.m, 0)


def _oracle_crt_local(self, Cblk, alpha_ptr, beta_ptr):
    mid = self.C_mid.reshape(-1)
    ldc = Cblk.stride(0) if Cblk.shape[0] > 1 else max(Cblk.shape[1], Cblk.stride(0))   # column-major block inside a larger C
    self.lib.oz2_invscal(self.dt, 0, self.N, self.m, self.n, mid.ctypes.data, ol._p(self.sftA), ol._p(self.sftB), alpha_ptr, beta_ptr,
                         C.c_void_p(Cblk.data_ptr()), ldc, 0)


OracleEngine.crt_local = _oracle_crt_local


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, N, fast, m, n, k, q, exchange="p2p"):
    os.environ["GEMMUL8_DIST_EXCHANGE"] = exchange
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import gemmul8_amd as g
        from gemmul8_amd import dist as gd
        rng = np.random.default_rng(42)
        A = (rng.random((m, k)) - 0.5) * np.exp(rng.standard_normal((m, k)))
        B = (rng.random((k, n)) - 0.5) * np.exp(rng.standard_normal((k, n)))
        C0 = rng.standard_normal((m, n))
        eng = OracleEngine(A, B, N, fast)
        plan = gd.ShardedGemm(g.D, g.INT8, m, n, k, N, fastmode=fast, engine=eng, alpha=-1.5, beta=1.5)
        Cmat = torch.from_numpy(np.ascontiguousarray(C0.T))  # (n, m): column-major view
        plan.run(None, None, Cmat)
        full = plan.gather_result(Cmat)
        if rank == 0:
            ref = ol.gemm(A, B, N, fastmode=fast, alpha=-1.5, beta=1.5, C0=C0)
            same = np.ascontiguousarray(full.numpy().T).tobytes() == np.ascontiguousarray(ref).tobytes()
            q.put((same, plan.t0, plan.t1, plan.c0, plan.c1))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _worker_cols(rank, world, port, N, fast, m, n, k, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import gemmul8_amd as g
        from gemmul8_amd import dist as gd
        rng = np.random.default_rng(43)
        A = (rng.random((m, k)) - 0.5) * np.exp(rng.standard_normal((m, k)))
        B = (rng.random((k, n)) - 0.5) * np.exp(rng.standard_normal((k, n)))
        C0 = rng.standard_normal((m, n))
        c0, c1 = gd.split_range(n, world, rank)
        eng = OracleEngine(A, B[:, c0:c1], N, fast) if c1 > c0 else None
        plan = gd.ColumnShardedGemm(g.D, g.INT8, m, n, k, N, fastmode=fast, engine=eng, alpha=-1.5, beta=1.5, mp=m)
        assert (plan.c0, plan.c1) == (c0, c1)
        Cmat = torch.from_numpy(np.ascontiguousarray(C0.T))
        plan.run(None, None, Cmat)
        full = plan.gather_result(Cmat)
        if rank == 0:
            ref = ol.gemm(A, B, N, fastmode=fast, alpha=-1.5, beta=1.5, C0=C0)
            q.put(np.ascontiguousarray(full.numpy().T).tobytes() == np.ascontiguousarray(ref).tobytes())
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _worker_blocks(rank, world, port, N, fast, m, n, k, q, grid):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import gemmul8_amd as g
        from gemmul8_amd import dist as gd
        rng = np.random.default_rng(44)
        A = (rng.random((m, k)) - 0.5) * np.exp(rng.standard_normal((m, k)))
        B = (rng.random((k, n)) - 0.5) * np.exp(rng.standard_normal((k, n)))
        C0 = rng.standard_normal((m, n))
        gr, gc = grid if grid else gd.block_grid(world)
        ri, cj = divmod(rank, gc)
        r0, r1 = gd.split_range(m, gr, ri)
        c0, c1 = gd.split_range(n, gc, cj)
        eng = OracleEngine(A[r0:r1], B[:, c0:c1], N, fast) if (r1 > r0 and c1 > c0) else None
        plan = gd.BlockShardedGemm(g.D, g.INT8, m, n, k, N, fastmode=fast, engine=eng, alpha=-1.5, beta=1.5, grid=grid)
        assert (plan.r0, plan.r1, plan.c0, plan.c1) == (r0, r1, c0, c1)
        Cmat = torch.from_numpy(np.ascontiguousarray(C0.T))
        plan.run(None, None, Cmat)
        full = plan.gather_result(Cmat)
        if rank == 0:
            ref = ol.gemm(A, B, N, fastmode=fast, alpha=-1.5, beta=1.5, C0=C0)
            q.put(np.ascontiguousarray(full.numpy().T).tobytes() == np.ascontiguousarray(ref).tobytes())
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,grid", [(2, None), (4, None), (4, (4, 1)), (3, None), (6, None)])
@pytest.mark.parametrize("N,fast,m,n", [(14, False, 19, 11), (9, True, 19, 11), (5, False, 3, 2)])
def test_block_sharded_gemm_matches_single_process(world, grid, N, fast, m, n):
    """Rows x columns block sharding (the default plan): one all_reduce(MAX) over the combined row/column bound vector;
    m = 3, n = 2 leaves ranks without a block that still have to take part in the collective."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    k = 37
    procs = [ctx.Process(target=_worker_blocks, args=(r, world, port, N, fast, m, n, k, q, grid)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    assert q.get(timeout=10), "block-sharded result differs from the single-process oracle"


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("N,fast,n", [(14, False, 11), (9, True, 11), (5, False, 2)])
def test_column_sharded_gemm_matches_single_process(world, N, fast, n):
    """Column-block sharding (the default plan): one all_reduce(MAX) of the row bounds; n = 2 with 3 ranks leaves a rank
    without columns that still has to take part in the collective."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    m, k = 19, 37
    procs = [ctx.Process(target=_worker_cols, args=(r, world, port, N, fast, m, n, k, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert q.get(timeout=10), "column-sharded result differs from the single-process oracle"


@pytest.mark.parametrize("world,exchange", [(2, "p2p"), (3, "p2p"), (2, "a2a"), (3, "a2a")])
@pytest.mark.parametrize("N,fast", [(14, False), (9, True), (2, False)])
def test_sharded_gemm_matches_single_process(world, exchange, N, fast):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    m, n, k = 19, 11, 37
    procs = [ctx.Process(target=_worker, args=(r, world, port, N, fast, m, n, k, q, exchange)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    same, t0, t1, c0, c1 = q.get(timeout=10)
    assert same, "sharded result differs from the single-process oracle"


def test_split_range_is_a_partition():
    from gemmul8_amd.dist import split_range
    for total in (0, 1, 2, 14, 15, 16, 20, 8192, 8191):
        for parts in (1, 2, 3, 4, 8):
            edges = [split_range(total, parts, i) for i in range(parts)]
            assert edges[0][0] == 0 and edges[-1][1] == total
            assert all(edges[i][1] == edges[i + 1][0] for i in range(parts - 1))
            sizes = [b - a for a, b in edges]
            assert max(sizes) - min(sizes) <= 1
