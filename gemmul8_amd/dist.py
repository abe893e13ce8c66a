"""Moduli-sharded multi-GPU emulated GEMM (one process per GPU, torch.distributed over RCCL/xGMI).

New work defined by BASELINE.json's north_star -- the reference has no multi-GPU code
(SURVEY.md 2.1, 8e).  The num_moduli residue pipelines are independent between the shift
computation and the CRT sum, so the path shards with exactly two small exchanges and one bulk one:

  rank r owns moduli [t0_r, t1_r) (contiguous, balanced) and output columns [c0_r, c1_r).
  1. bounds   (accurate mode) every rank extracts the 7-bit bound planes, runs the bound GEMM only
              on ITS column block, then ONE all-reduce(MAX) over int32[mp + pad(n)] gives every rank
              the full row/column maxima  ->  identical shifts everywhere.  (fast mode: no exchange,
              shifts are recomputed redundantly from A and B, which are replicated.)
  2. finish   shifts + residue planes of A and B for the rank's moduli only.
  3. lowprec  the rank's INT8 MFMA GEMMs with the requantise epilogue -> its C_mid planes (m x n int8).
  4. exchange residue all-to-all: plane t, column block s goes to rank s (point-to-point over
              xGMI, all 7 links busy; 7/8 * N/G * m*n bytes out per rank -- 8x less than exchanging
              FP64 partial sums) into a [N][cols_r][mp] buffer.
  5. crt      reference-order CRT accumulation on the rank's columns -> C[:, c0_r:c1_r].
Integer intermediates and the final C are bit-identical to the single-GPU result for every G,
because each output element still sees all N residues in the order t = 0..N-1.

The compute engine is pluggable so that the sharding/exchange logic can be tested on CPU with the
gloo backend (tests plug the CPU oracle in); the product engine is HipEngine (C ABI, device memory).
"""
import ctypes as C

import torch
import torch.distributed as dist

import gemmul8_amd as g


def _ld(t):
    """Leading dimension (elements between consecutive columns) of a column-major matrix held as a (cols, rows) tensor or as
    a row-sliced view of one."""
    return t.stride(0) if t.shape[0] > 1 else max(t.shape[1], t.stride(0))


def block_grid(world):
    """(Gr, Gc) with Gr * Gc == world and Gr >= Gc as square as possible: 2 -> 2x1, 4 -> 2x2, 8 -> 4x2.  The row side gets the
    larger factor because the per-operand scaling work of A (row-strided for op N: amax pass + LDS-staged extract/quantise,
    0.62 ms at 8192^2) costs more than that of B (0.44 ms), and it is divided by Gr."""
    gc = 1
    for d in range(1, int(world ** 0.5) + 1):
        if world % d == 0:
            gc = d
    return world // gc, gc


def split_range(total, parts, idx):
    """Balanced contiguous split: the first (total % parts) pieces get one extra."""
    q, r = divmod(total, parts)
    b = idx * q + min(idx, r)
    return b, b + q + (1 if idx < r else 0)


class HipEngine:
    """Phase calls through the C ABI on one GPU; all buffers are views of one torch uint8 workspace."""

    def __init__(self, dtype_code, backend, m, n, k, N, fastmode, device, opA="N", opB="N"):
        self.lib = g.lib()
        self.dt, self.be, self.m, self.n, self.k, self.N, self.fast = dtype_code, backend, m, n, k, N, int(fastmode)
        self.opA, self.opB = g.OPS[opA], g.OPS[opB]
        self.device = device
        cplx = dtype_code >= 2
        tot, _, _ = g.work_size(cplx, backend, m, n, k, N)
        self.work = torch.empty(tot, dtype=torch.uint8, device=device)
        self.L = g.Layout()
        g.check(self.lib.gemmul8_get_layout(dtype_code, backend, m, n, k, N, self.work.data_ptr(), None, None, 0, 0, C.byref(self.L)))
        self.mid_bytes = (1 if backend == g.INT8 else 2) * (2 if cplx else 1)
        self.elem_bytes = {g.S: 4, g.D: 8, g.Cx: 8, g.Z: 16}[dtype_code]
        self.mp = self.L.mp
        self.np_ = (n + 255) // 256 * 256

    def _view(self, ptr, nbytes):
        off = ptr - self.work.data_ptr()
        return self.work[off:off + nbytes]

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def maxima(self):
        """int32 tensor [mp + pad(n)] = rowmax | colmax (views of the workspace scratch)."""
        return self._view(self.L.scratch, 4 * (self.mp + self.np_)).view(torch.int32)

    def bounds(self, A, B, c0, c1):
        g.check(self.lib.gemmul8_scale_bounds(self._stream(), self.dt, self.be, self.opA, self.opB, self.m, self.n, self.k, A.data_ptr(),
                                              _ld(A), B.data_ptr(), _ld(B), self.N, c0, c1, C.byref(self.L), 0, 0), "scale_bounds")

    def finish(self, A, B, t0, t1):
        g.check(self.lib.gemmul8_scale_finish(self._stream(), self.dt, self.be, self.opA, self.opB, self.m, self.n, self.k, A.data_ptr(),
                                              _ld(A), B.data_ptr(), _ld(B), self.N, self.fast, t0, t1, C.byref(self.L), 0, 0), "scale_finish")

    def lowprec(self, t0, t1):
        g.check(self.lib.gemmul8_lowprec_gemm(self._stream(), self.dt, self.be, self.m, self.n, self.k, self.N, t0, t1, C.byref(self.L)), "lowprec_gemm")

    def plane_block(self, t, c0, c1):
        """uint8 view of residue plane t, columns [c0, c1): contiguous (c1-c0)*mp*mid_bytes bytes."""
        base = self.L.C_mid + (t * self.L.sizeC + c0 * self.mp) * self.mid_bytes
        return self._view(base, (c1 - c0) * self.mp * self.mid_bytes)

    def new_recv(self, ncols):
        return torch.empty(self.N * ncols * self.mp * self.mid_bytes, dtype=torch.uint8, device=self.device)

    def sft_ptrs(self):
        return self.L.sftA, self.L.sftB

    def crt_local(self, Cblk, alpha_ptr, beta_ptr):
        """CRT of the engine's own C_mid planes into the (n, ld) tensor Cblk (column sharding: the engine IS the block)."""
        sA, sB = self.sft_ptrs()
        ldc = _ld(Cblk)
        g.check(self.lib.gemmul8_crt(self._stream(), self.dt, self.be, self.N, self.m, self.n, self.L.C_mid, self.mp, self.L.sizeC, sA, sB,
                                     alpha_ptr, beta_ptr, Cblk.data_ptr(), ldc), "crt")

    def crt(self, recv, c0, c1, Cmat, alpha_ptr, beta_ptr):
        ncols = c1 - c0
        if ncols == 0:
            return
        sA, sB = self.sft_ptrs()
        ldc = Cmat.shape[1]
        g.check(self.lib.gemmul8_crt(self._stream(), self.dt, self.be, self.N, self.m, ncols, recv.data_ptr(), self.mp, ncols * self.mp,
                                     sA, sB + 2 * c0, alpha_ptr, beta_ptr, Cmat.data_ptr() + c0 * ldc * self.elem_bytes, ldc), "crt")


class ColumnShardedGemm:
    """C[:, cols_r] = alpha*op(A)*op(B[:, cols_r]) + beta*C[:, cols_r]: the output COLUMNS are sharded, every rank runs all
    num_moduli residue pipelines on its column block.

    Column blocks are independent stripes of the path except for one thing: the accurate-mode row shifts of A depend on the
    row maxima of the bound product over ALL columns, so the ranks all-reduce(MAX) int32[mp] between the bound phase and
    the quantise phase (fast mode: no collective at all).  Compared with moduli sharding there is no bulk exchange
    (7/8 * N/G * m*n residue bytes per rank) and no imbalance when num_moduli is not a multiple of the GPU count
    (14 moduli on 8 GPUs: 2,2,2,2,2,2,1,1); the price is that every rank quantises all N planes of A.  Results are
    bit-identical to the single-GPU call for every world size (same shifts, same per-element arithmetic).
    A, B replicated, C column-sharded, as for ShardedGemm."""

    def __init__(self, dtype_code, backend, m, n, k, N, fastmode=False, device=None, group=None, engine=None, alpha=1.0, beta=0.0,
                 mp=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.m, self.n, self.k, self.N, self.fast = m, n, k, N, fastmode
        self.c0, self.c1 = split_range(n, self.world, self.rank)
        self.ncols = self.c1 - self.c0
        self.my_planes = N
        self.dt = dtype_code
        self.elem_bytes = {g.S: 4, g.D: 8, g.Cx: 8, g.Z: 16}[dtype_code]
        # engine for the (m, ncols, k) sub-problem; a rank without columns (n < world) only takes part in the all-reduce
        self.eng = engine if engine is not None else (HipEngine(dtype_code, backend, m, self.ncols, k, N, fastmode, device) if self.ncols else None)
        # length of the row-bound vector every rank reduces (HipEngine pads rows to 256; a plugged-in engine may not)
        self.mp = self.eng.mp if self.eng is not None else (mp if mp is not None else (m + 255) // 256 * 256)
        self.device = device
        import numpy as np
        np_dt = {g.S: np.float32, g.D: np.float64, g.Cx: np.complex64, g.Z: np.complex128}[dtype_code]
        self._alpha = np.array([alpha], dtype=np_dt)
        self._beta = np.array([beta], dtype=np_dt)

    def run(self, A, B, Cmat, record_gemm_events=False):
        eng = self.eng
        Bblk = B[self.c0:self.c1] if B is not None else None      # tensor rows = matrix columns
        Cblk = Cmat[self.c0:self.c1]
        if not self.fast:
            if eng is not None:
                eng.bounds(A, Bblk, 0, self.ncols)
                rowmax = eng.maxima()[:self.mp]
            else:
                rowmax = torch.zeros(self.mp, dtype=torch.int32, device=self.device if self.device is not None else "cpu")
            if self.world > 1:
                stage = rowmax.is_cuda and dist.get_backend(self.group) == "gloo"  # single-GPU multi-rank smoke test only
                if stage:
                    host = rowmax.cpu()
                    dist.all_reduce(host, op=dist.ReduceOp.MAX, group=self.group)
                    rowmax.copy_(host)
                else:
                    dist.all_reduce(rowmax, op=dist.ReduceOp.MAX, group=self.group)
        if eng is None:
            return None
        eng.finish(A, Bblk, 0, self.N)
        ev = None
        if record_gemm_events and torch.cuda.is_available() and Cmat.is_cuda:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            eng.lowprec(0, self.N)
            e1.record()
            ev = (e0, e1)
        else:
            eng.lowprec(0, self.N)
        eng.crt_local(Cblk, self._alpha.ctypes.data, self._beta.ctypes.data)
        return ev

    gather_result = None  # bound below (same assembly as ShardedGemm)


class BlockShardedGemm:
    """C[rows_i, cols_j] = alpha*A[rows_i, :]*B[:, cols_j] + beta*C[rows_i, cols_j] on a Gr x Gc grid of ranks (op N/N).

    Rank (i, j) runs the ordinary single-GPU phase calls on its (m/Gr, n/Gc, k) sub-problem: it reads only its row block
    of A and its column block of B, so the replicated scaling work of the column plan (every rank quantising all of A)
    drops by Gr, and the INT8 GEMM work is 1/G for any num_moduli.  The only coupling is the accurate mode's bound
    maxima: a row's shift needs the row maximum over ALL columns and a column's over ALL rows, so every rank writes its
    partial maxima into one zero-filled vector int32[M + N] (its rows, its columns) and ONE all_reduce(MAX) over all ranks
    completes both; fast mode needs no collective.  Bit-identical to the single-GPU call for every grid."""

    def __init__(self, dtype_code, backend, m, n, k, N, fastmode=False, device=None, group=None, engine=None, alpha=1.0, beta=0.0,
                 grid=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.gr, self.gc = grid if grid is not None else block_grid(self.world)
        assert self.gr * self.gc == self.world
        self.m, self.n, self.k, self.N, self.fast = m, n, k, N, fastmode
        self.ri, self.cj = divmod(self.rank, self.gc)
        self.r0, self.r1 = split_range(m, self.gr, self.ri)
        self.c0, self.c1 = split_range(n, self.gc, self.cj)
        self.nrows, self.ncols = self.r1 - self.r0, self.c1 - self.c0
        self.my_planes = N
        self.device = device
        empty = self.nrows == 0 or self.ncols == 0
        self.eng = engine if engine is not None else (None if empty else HipEngine(dtype_code, backend, self.nrows, self.ncols, k, N, fastmode, device))
        import numpy as np
        np_dt = {g.S: np.float32, g.D: np.float64, g.Cx: np.complex64, g.Z: np.complex128}[dtype_code]
        self._alpha = np.array([alpha], dtype=np_dt)
        self._beta = np.array([beta], dtype=np_dt)
        self._mx = None

    def _blocks(self, A, B, Cmat):
        Ablk = A[:, self.r0:self.r1] if A is not None else None      # (k, rows): column-major rows x k, ld = m
        Bblk = B[self.c0:self.c1] if B is not None else None         # (cols, k): column-major k x cols
        Cblk = Cmat[self.c0:self.c1, self.r0:self.r1]                # (cols, rows): column-major rows x cols, ld = m
        return Ablk, Bblk, Cblk

    def run(self, A, B, Cmat, record_gemm_events=False):
        eng = self.eng
        Ablk, Bblk, Cblk = self._blocks(A, B, Cmat)
        if not self.fast:
            if self.world > 1:
                dev = Cmat.device
                if self._mx is None:
                    self._mx = torch.zeros(self.m + self.n, dtype=torch.int32, device=dev)
                mx = self._mx
                mx.zero_()
            if eng is not None:
                eng.bounds(Ablk, Bblk, 0, self.ncols)
                loc = eng.maxima()
            if self.world > 1:
                if eng is not None:
                    mx[self.r0:self.r1].copy_(loc[:self.nrows])
                    mx[self.m + self.c0:self.m + self.c1].copy_(loc[eng.mp:eng.mp + self.ncols])
                stage = mx.is_cuda and dist.get_backend(self.group) == "gloo"  # single-GPU multi-rank smoke test only
                if stage:
                    host = mx.cpu()
                    dist.all_reduce(host, op=dist.ReduceOp.MAX, group=self.group)
                    mx.copy_(host)
                else:
                    dist.all_reduce(mx, op=dist.ReduceOp.MAX, group=self.group)
                if eng is not None:
                    loc[:self.nrows].copy_(mx[self.r0:self.r1])
                    loc[eng.mp:eng.mp + self.ncols].copy_(mx[self.m + self.c0:self.m + self.c1])
        if eng is None:
            return None
        eng.finish(Ablk, Bblk, 0, self.N)
        ev = None
        if record_gemm_events and torch.cuda.is_available() and Cmat.is_cuda:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            eng.lowprec(0, self.N)
            e1.record()
            ev = (e0, e1)
        else:
            eng.lowprec(0, self.N)
        eng.crt_local(Cblk, self._alpha.ctypes.data, self._beta.ctypes.data)
        return ev

    def gather_result(self, Cmat):
        """All ranks: assemble the full C from the blocks (verification / callers that need it)."""
        if self.world == 1:
            return Cmat
        out = Cmat.clone()
        for s in range(self.world):
            si, sj = divmod(s, self.gc)
            r0, r1 = split_range(self.m, self.gr, si)
            c0, c1 = split_range(self.n, self.gc, sj)
            if r1 > r0 and c1 > c0:
                blk = Cmat[c0:c1, r0:r1].contiguous()
                src = s if self.group is None else dist.get_global_rank(self.group, s)
                dist.broadcast(blk, src=src, group=self.group)
                out[c0:c1, r0:r1] = blk
        return out


class ShardedGemm:
    """C[:, cols_r] = alpha*op(A)*op(B) + beta*C[:, cols_r] with the moduli sharded over the process group.

    A, B: replicated on every rank (column-major as tensors of shape (cols, rows)); C: every rank
    updates only its own column block of its local C (gather_result() assembles the full matrix)."""

    def __init__(self, dtype_code, backend, m, n, k, N, fastmode=False, device=None, group=None, engine=None, alpha=1.0, beta=0.0):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.m, self.n, self.k, self.N, self.fast = m, n, k, N, fastmode
        self.t0, self.t1 = split_range(N, self.world, self.rank)
        self.c0, self.c1 = split_range(n, self.world, self.rank)
        self.my_planes = self.t1 - self.t0
        self.eng = engine if engine is not None else HipEngine(dtype_code, backend, m, n, k, N, fastmode, device)
        self.recv = self.eng.new_recv(self.c1 - self.c0)
        import os
        self.exchange_mode = os.environ.get("GEMMUL8_DIST_EXCHANGE", "p2p")  # "p2p": batch_isend_irecv, "a2a": all_to_all_single
        import numpy as np
        np_dt = {g.S: np.float32, g.D: np.float64, g.Cx: np.complex64, g.Z: np.complex128}[dtype_code]
        self._alpha = np.array([alpha], dtype=np_dt)
        self._beta = np.array([beta], dtype=np_dt)

    def _recv_slot(self, t):
        ncols = self.c1 - self.c0
        sz = ncols * self.eng.mp * self.eng.mid_bytes
        return self.recv[t * sz:(t + 1) * sz]

    def exchange_a2a(self):
        """The same exchange as ONE all_to_all_single (GEMMUL8_DIST_EXCHANGE=a2a): the send buffer is packed
        [dest s][my planes t][cols of s][mp]; because ranks own contiguous moduli ranges in rank order, the received
        buffer [source r][planes of r][my cols][mp] IS the [t = 0..N-1][my cols][mp] layout the CRT reads."""
        eng = self.eng
        unit = eng.mp * eng.mid_bytes
        in_splits, chunks = [], []
        for s in range(self.world):
            sc0, sc1 = split_range(self.n, self.world, s)
            in_splits.append(self.my_planes * (sc1 - sc0) * unit)
            for t in range(self.t0, self.t1):
                if sc1 > sc0:
                    chunks.append(eng.plane_block(t, sc0, sc1))
        out_splits = []
        for s in range(self.world):
            st0, st1 = split_range(self.N, self.world, s)
            out_splits.append((st1 - st0) * (self.c1 - self.c0) * unit)
        send = torch.cat(chunks) if chunks else self.recv.new_empty(0)
        stage = self.recv.is_cuda and dist.get_backend(self.group) == "gloo"
        if stage:
            host = torch.empty(self.recv.shape, dtype=self.recv.dtype)
            dist.all_to_all_single(host, send.cpu(), out_splits, in_splits, group=self.group)
            self.recv.copy_(host)
        else:
            dist.all_to_all_single(self.recv, send, out_splits, in_splits, group=self.group)

    def exchange(self):
        """Residue all-to-all: my planes' column block s -> rank s; planes of rank s for my columns <- rank s."""
        if self.world > 1 and self.exchange_mode == "a2a":
            return self.exchange_a2a()
        ops = []
        # gloo cannot send/recv device tensors: stage through host memory (only used by the single-GPU
        # 2-rank correctness test; the product path is NCCL/RCCL with device buffers)
        stage = self.recv.is_cuda and dist.get_backend(self.group) == "gloo"
        staged = []
        for s in range(self.world):
            sc0, sc1 = split_range(self.n, self.world, s)
            st0, st1 = split_range(self.N, self.world, s)
            if s == self.rank:
                for t in range(self.t0, self.t1):
                    self._recv_slot(t).copy_(self.eng.plane_block(t, self.c0, self.c1))
                continue
            peer = s if self.group is None else dist.get_global_rank(self.group, s)
            if sc1 > sc0:
                for t in range(self.t0, self.t1):
                    blk = self.eng.plane_block(t, sc0, sc1)
                    ops.append(dist.P2POp(dist.isend, blk.cpu() if stage else blk, peer, self.group))
            if self.c1 > self.c0:
                for t in range(st0, st1):
                    slot = self._recv_slot(t)
                    if stage:
                        host = torch.empty(slot.shape, dtype=slot.dtype)
                        staged.append((slot, host))
                        slot = host
                    ops.append(dist.P2POp(dist.irecv, slot, peer, self.group))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        for slot, host in staged:
            slot.copy_(host)

    def run(self, A, B, Cmat, record_gemm_events=False):
        eng = self.eng
        if not self.fast:
            eng.bounds(A, B, self.c0, self.c1)
            if self.world > 1:
                dist.all_reduce(eng.maxima(), op=dist.ReduceOp.MAX, group=self.group)
        eng.finish(A, B, self.t0, self.t1)
        ev = None
        if record_gemm_events and self.my_planes > 0 and torch.cuda.is_available():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            eng.lowprec(self.t0, self.t1)
            e1.record()
            ev = (e0, e1)
        else:
            eng.lowprec(self.t0, self.t1)
        self.exchange()
        eng.crt(self.recv, self.c0, self.c1, Cmat, self._alpha.ctypes.data, self._beta.ctypes.data)
        return ev

    def gather_result(self, Cmat):
        """All ranks: assemble the full C (tensor (n, ld)) from the column blocks (verification / callers that need it)."""
        if self.world == 1:
            return Cmat
        out = Cmat.clone()
        for s in range(self.world):
            sc0, sc1 = split_range(self.n, self.world, s)
            if sc1 > sc0:
                blk = out[sc0:sc1].contiguous() if s != self.rank else Cmat[sc0:sc1].contiguous()
                src = s if self.group is None else dist.get_global_rank(self.group, s)
                dist.broadcast(blk, src=src, group=self.group)
                out[sc0:sc1] = blk
        return out


ColumnShardedGemm.gather_result = ShardedGemm.gather_result


def make_plan(dtype_code, backend, m, n, k, N, **kw):
    """The multi-GPU plan bench.py and callers use: GEMMUL8_DIST_SHARD=blocks (default) | columns | moduli."""
    import os
    mode = os.environ.get("GEMMUL8_DIST_SHARD", "blocks")
    cls = {"blocks": BlockShardedGemm, "columns": ColumnShardedGemm, "moduli": ShardedGemm}[mode]
    return cls(dtype_code, backend, m, n, k, N, **kw)
