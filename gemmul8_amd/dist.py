"""Multi-GPU emulated GEMM: ctypes caller of the C ABI in include/gemmul8_dist.h (gemmul8_amd/csrc/oz2_dist.cpp).

All sharding logic -- the three plans (output blocks on a rank grid; moduli with an INT8 residue exchange; moduli with the
FP64 partial-sum reduce-scatter that BASELINE.json's north_star names), the partition arithmetic, the RCCL transport -- lives
in C++ behind that boundary, so that a C++ caller of gemmul8::gemm or an application under the LD_PRELOAD hook reaches the
same code.  This module only
  * builds the RCCL communicator of the product path (the 128-byte ncclUniqueId travels over torch.distributed's store),
  * offers `TorchTransport`, a gemmul8_comm table implemented with torch.distributed calls through ctypes callbacks -- the
    TEST transport: gloo on CPU (tests/test_dist_cpu.py drives the C++ plans at world sizes 2..8 with host memory) and
    host-staged gloo for several ranks sharing the one GPU of a test box,
  * offers `TorchNcclTransport`, the same table over torch's own nccl (= RCCL) process group on the device buffers -- bench.py's
    second choice on a multi-GPU node when the library's own communicator cannot be brought up,
  * wraps a plan as `DistGemm` for bench.py and the tests.

Placement: A and B replicated on every rank, C full-size on every rank, each rank updating the block it owns
(`DistGemm.owned_block`); `gather_result` assembles the whole matrix everywhere.
"""
import ctypes as C
import os

import gemmul8_amd as g

BLOCKS, MODULI, MODULI_FP64SUM = 0, 1, 2
PLAN_CODES = {"blocks": BLOCKS, "moduli": MODULI, "fp64sum": MODULI_FP64SUM}


class P2POp(C.Structure):
    _fields_ = [("buf", C.c_void_p), ("bytes", C.c_size_t), ("peer", C.c_int), ("is_send", C.c_int)]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
SENDRECV_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(P2POp), C.c_void_p)
REDSCAT_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
DESTROY_FN = C.CFUNCTYPE(None, C.c_void_p)


class Comm(C.Structure):
    """Mirror of struct gemmul8_comm."""
    _fields_ = [("ctx", C.c_void_p), ("rank", C.c_int), ("world", C.c_int), ("allreduce_max_i32", ALLREDUCE_FN),
                ("sendrecv", SENDRECV_FN), ("reduce_scatter_sum_f64", REDSCAT_FN), ("destroy", DESTROY_FN)]


_LAYOUT_P = C.POINTER(g.Layout)
ENGINE_FIELDS = [
    ("alloc", C.CFUNCTYPE(C.c_void_p, C.c_size_t)),
    ("release", C.CFUNCTYPE(None, C.c_void_p)),
    ("zero", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t, C.c_void_p)),
    ("copy", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)),
    ("copy2d", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p)),
    ("scale_bounds", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p,
                                 C.c_size_t, C.c_void_p, C.c_size_t, C.c_uint, C.c_size_t, C.c_size_t, _LAYOUT_P, C.c_int, C.c_int)),
    ("scale_finish", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p,
                                 C.c_size_t, C.c_void_p, C.c_size_t, C.c_uint, C.c_int, C.c_uint, C.c_uint, _LAYOUT_P, C.c_int, C.c_int)),
    ("lowprec_gemm", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.c_uint, C.c_uint, C.c_uint,
                                 _LAYOUT_P)),
    ("crt", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_uint, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t,
                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)),
    ("crt_partial", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_uint, C.c_uint, C.c_uint, C.c_size_t, C.c_size_t, C.c_void_p,
                                C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t)),
    ("crt_finish", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_uint, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t,
                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)),
    ("add_f64", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)),   # ABI 7: dst += src on doubles (pipelined fp64sum)
]


class Engine(C.Structure):
    """Mirror of struct gemmul8_dist_engine (NULL = the HIP engine; tests fill it with CPU-oracle callbacks)."""
    _fields_ = ENGINE_FIELDS


_bound = False


def _lib():
    global _bound
    L = g.lib()
    if not _bound:
        L.gemmul8_comm_rccl_unique_id.restype = C.c_int
        L.gemmul8_comm_rccl_unique_id.argtypes = [C.c_void_p]
        L.gemmul8_comm_rccl_create.restype = C.c_int
        L.gemmul8_comm_rccl_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.POINTER(Comm))]
        L.gemmul8_comm_rccl_id_from_env.restype = C.c_int
        L.gemmul8_comm_rccl_id_from_env.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.gemmul8_comm_rccl_from_env.restype = C.c_int
        L.gemmul8_comm_rccl_from_env.argtypes = [C.POINTER(C.POINTER(Comm))]
        L.gemmul8_comm_destroy.restype = None
        L.gemmul8_comm_destroy.argtypes = [C.POINTER(Comm)]
        L.gemmul8_dist_create.restype = C.c_int
        L.gemmul8_dist_create.argtypes = [C.POINTER(Comm), C.POINTER(Engine), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.c_size_t, C.c_size_t, C.c_size_t, C.c_uint, C.c_int, C.POINTER(C.c_void_p)]
        L.gemmul8_dist_gemm.restype = C.c_int
        L.gemmul8_dist_gemm.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p,
                                        C.c_void_p, C.c_size_t]
        L.gemmul8_dist_owned_block.restype = C.c_int
        L.gemmul8_dist_owned_block.argtypes = [C.c_void_p, C.c_int] + [C.POINTER(C.c_size_t)] * 4
        L.gemmul8_dist_my_work.restype = C.c_int
        L.gemmul8_dist_my_work.argtypes = [C.c_void_p, C.POINTER(C.c_uint), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        L.gemmul8_dist_allgather_c.restype = C.c_int
        L.gemmul8_dist_allgather_c.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        L.gemmul8_dist_set_events.restype = C.c_int
        L.gemmul8_dist_set_events.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.gemmul8_comm_rccl_count.restype = C.c_int
        L.gemmul8_comm_rccl_count.argtypes = [C.POINTER(Comm), C.POINTER(C.c_int)]
        L.gemmul8_dist_set_exchange_events.restype = C.c_int
        L.gemmul8_dist_set_exchange_events.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        L.gemmul8_dist_exchange_bytes.restype = C.c_int
        L.gemmul8_dist_exchange_bytes.argtypes = [C.c_void_p] + [C.POINTER(C.c_size_t)] * 3
        L.gemmul8_dist_workspace_bytes.restype = C.c_size_t
        L.gemmul8_dist_workspace_bytes.argtypes = [C.c_void_p]
        L.gemmul8_dist_destroy.restype = None
        L.gemmul8_dist_destroy.argtypes = [C.c_void_p]
        _bound = True
    return L


class RcclComm:
    """The product transport: an RCCL communicator created inside libgemmul8.so (ncclCommInitRank on the current device); the
    ncclUniqueId is generated on rank 0 and handed to the other ranks through torch.distributed."""

    def __init__(self, group=None):
        import torch.distributed as dist
        L = _lib()
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        ident = C.create_string_buffer(128)
        if rank == 0:
            g.check(L.gemmul8_comm_rccl_unique_id(ident), "gemmul8_comm_rccl_unique_id")
        raw = ident.raw
        if world > 1:
            # plain tensor broadcast on the group's own device type (the most exercised torch.distributed path: no pickling, no
            # object collectives); nccl groups need a device tensor on the current device, gloo groups a host tensor
            import torch
            on_dev = dist.get_backend(group) == "nccl"
            t = torch.tensor(list(raw), dtype=torch.uint8, device=torch.device("cuda", torch.cuda.current_device()) if on_dev else "cpu")
            src = 0 if group is None else dist.get_global_rank(group, 0)
            dist.broadcast(t, src=src, group=group)
            raw = bytes(t.cpu().tolist())
        self.ptr = C.POINTER(Comm)()
        g.check(L.gemmul8_comm_rccl_create(C.create_string_buffer(raw, 128), rank, world, C.byref(self.ptr)), "gemmul8_comm_rccl_create")
        self.rank, self.world = rank, world

    def rccl_ranks(self):
        """Size of the communicator as RCCL reports it (ncclCommCount)."""
        n = C.c_int(-1)
        g.check(_lib().gemmul8_comm_rccl_count(self.ptr, C.byref(n)), "gemmul8_comm_rccl_count")
        return n.value

    def close(self):
        if self.ptr:
            _lib().gemmul8_comm_destroy(self.ptr)
            self.ptr = None


class TorchTransport:
    """gemmul8_comm implemented with torch.distributed calls (ctypes callbacks).  TEST transport only: `device=False` works on
    host buffers (CPU tests of the C++ plans over gloo); `device=True` stages device buffers through host memory so that
    several gloo ranks can share one GPU.  The product path is RcclComm."""

    def __init__(self, group=None, device=False):
        import torch.distributed as dist
        self.dist, self.group, self.device = dist, group, device
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self._hip = C.CDLL(g._bind_hip_runtime()) if device else None
        self._keep = [ALLREDUCE_FN(self._allreduce), SENDRECV_FN(self._sendrecv), REDSCAT_FN(self._redscat), DESTROY_FN(lambda ctx: None)]
        self.struct = Comm(None, self.rank, self.world, *self._keep)
        self.ptr = C.pointer(self.struct)

    def rccl_ranks(self):
        return -1  # not an RCCL transport

    def close(self):
        pass

    # -- buffers: host pointer -> numpy view; device pointer -> host copy (and back)
    def _fetch(self, ptr, nbytes, stream):
        import numpy as np
        if not self.device:
            return np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(ptr))
        self._hip.hipStreamSynchronize(C.c_void_p(stream))
        host = np.empty(nbytes, np.uint8)
        rc = self._hip.hipMemcpy(host.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), C.c_size_t(nbytes), 2)
        assert rc == 0, rc
        return host

    def _store(self, ptr, host):
        if self.device:
            rc = self._hip.hipMemcpy(C.c_void_p(ptr), host.ctypes.data_as(C.c_void_p), C.c_size_t(host.nbytes), 1)
            assert rc == 0, rc

    def _peer(self, r):
        return r if self.group is None else self.dist.get_global_rank(self.group, r)

    def _allreduce(self, ctx, buf, count, stream):
        import numpy as np
        import torch
        try:
            host = self._fetch(buf, 4 * count, stream)
            t = torch.from_numpy(host.view(np.int32))
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
            self._store(buf, host)
            return 0
        except Exception as e:  # an exception must not unwind through the C++ caller
            print("TorchTransport.allreduce failed:", e)
            return 1

    def _sendrecv(self, ctx, nops, ops, stream):
        import torch
        try:
            work, back = [], []
            for i in range(nops):
                op = ops[i]
                if op.bytes == 0:
                    continue
                host = self._fetch(op.buf, op.bytes, stream)
                t = torch.from_numpy(host)
                if op.is_send:
                    work.append(self.dist.P2POp(self.dist.isend, t, self._peer(op.peer), self.group))
                else:
                    work.append(self.dist.P2POp(self.dist.irecv, t, self._peer(op.peer), self.group))
                    back.append((op.buf, host))
            if work:
                for w in self.dist.batch_isend_irecv(work):
                    w.wait()
            for ptr, host in back:
                self._store(ptr, host)
            return 0
        except Exception as e:
            print("TorchTransport.sendrecv failed:", e)
            return 1

    def _redscat(self, ctx, send, recv, recv_count, stream):
        import numpy as np
        import torch
        try:
            src = self._fetch(send, 8 * recv_count * self.world, stream).view(np.float64)
            # gloo has no reduce_scatter: all_reduce the whole buffer and keep this rank's block.  The order in which a real
            # reduce-scatter adds the per-rank partials is the transport's; for two ranks it is unique (a + b == b + a).
            t = torch.from_numpy(src.copy())
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
            mine = t.numpy()[self.rank * recv_count:(self.rank + 1) * recv_count].copy()
            if self.device:
                self._store(recv, mine.view(np.uint8))
            else:
                C.memmove(recv, mine.ctypes.data, mine.nbytes)
            return 0
        except Exception as e:
            print("TorchTransport.reduce_scatter failed:", e)
            return 1


class _DevView:
    """A raw device pointer as a __cuda_array_interface__ object (version 2: no stream hand-shake), so that torch can wrap it."""

    def __init__(self, ptr, count, typestr):
        self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": typestr, "data": (int(ptr), False), "version": 2}


class TorchNcclTransport:
    """gemmul8_comm over an EXISTING torch.distributed nccl (= RCCL) process group, on the device buffers themselves: no host
    staging, the bytes travel over xGMI exactly as with RcclComm, but inside the communicator torch already brought up.
    bench.py's SECOND choice on a multi-GPU node: taken only when the library's own communicator (RcclComm, ncclCommInitRank in
    libgemmul8.so) cannot be created or fails its self-test, so that a first 8-GPU lease still returns a measured line; the JSON
    says which transport ran.  The collectives are issued on the stream the plan passes (wrapped as an ExternalStream when it is
    not torch's current one); torch's process group orders them after that stream's work and the stream after them."""

    def __init__(self, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.dev = torch.device("cuda", torch.cuda.current_device())
        self._keep = [ALLREDUCE_FN(self._allreduce), SENDRECV_FN(self._sendrecv), REDSCAT_FN(self._redscat), DESTROY_FN(lambda ctx: None)]
        self.struct = Comm(None, self.rank, self.world, *self._keep)
        self.ptr = C.pointer(self.struct)

    def rccl_ranks(self):
        return self.world  # torch's communicator: its size is the group's

    def close(self):
        pass

    def _peer(self, r):
        return r if self.group is None else self.dist.get_global_rank(self.group, r)

    def _view(self, ptr, count, typestr):
        return self.torch.as_tensor(_DevView(ptr, count, typestr), device=self.dev)

    def _on(self, stream):
        """Context that makes `stream` (a raw hipStream_t, None / 0 = the null stream) torch's current stream."""
        t = self.torch
        raw = int(stream or 0)
        if raw == t.cuda.current_stream(self.dev).cuda_stream:
            import contextlib
            return contextlib.nullcontext()
        return t.cuda.stream(t.cuda.ExternalStream(raw, device=self.dev) if raw else t.cuda.default_stream(self.dev))

    def _allreduce(self, ctx, buf, count, stream):
        try:
            with self._on(stream):
                self.dist.all_reduce(self._view(buf, count, "<i4"), op=self.dist.ReduceOp.MAX, group=self.group)
            return 0
        except Exception as e:  # an exception must not unwind through the C++ caller
            print("TorchNcclTransport.allreduce failed:", e)
            return 1

    def _sendrecv(self, ctx, nops, ops, stream):
        try:
            work = []
            for i in range(nops):
                op = ops[i]
                if op.bytes == 0:
                    continue
                t = self._view(op.buf, op.bytes, "|u1")
                work.append(self.dist.P2POp(self.dist.isend if op.is_send else self.dist.irecv, t, self._peer(op.peer), self.group))
            if work:
                with self._on(stream):
                    for w in self.dist.batch_isend_irecv(work):
                        w.wait()   # stream-level wait for nccl work: the host does not block
            return 0
        except Exception as e:
            print("TorchNcclTransport.sendrecv failed:", e)
            return 1

    def _redscat(self, ctx, send, recv, recv_count, stream):
        try:
            with self._on(stream):
                self.dist.reduce_scatter_tensor(self._view(recv, recv_count, "<f8"), self._view(send, recv_count * self.world, "<f8"),
                                                op=self.dist.ReduceOp.SUM, group=self.group)
            return 0
        except Exception as e:
            print("TorchNcclTransport.reduce_scatter failed:", e)
            return 1


def selftest(comm, device, stream=None, expect_rccl=True):
    """First-contact check of a transport (RcclComm or TorchTransport) before anything is timed on it: communicator size, one
    all-reduce(MAX) on int32, one grouped send/recv ring, one reduce-scatter(sum) on FP64 -- each compared with host arithmetic.
    Returns (ok, one-line diagnosis).  Every rank must call it (the three operations are collective)."""
    import numpy as np
    import torch
    rank, world = comm.rank, comm.world
    on_gpu = torch.device(device).type == "cuda"
    st = stream if stream is not None else (torch.cuda.current_stream(device).cuda_stream if on_gpu else None)
    sync = (lambda: torch.cuda.synchronize(device)) if on_gpu else (lambda: None)   # host buffers: the CPU test transport completes before returning
    c = comm.ptr.contents
    # The three operations are collectives: a rank that returned after a failed check would leave the others blocked in the next one
    # (and bench.py's agreement step would never be reached).  Every rank therefore runs ALL of them and reports the first failure.
    fails = []
    try:
        if expect_rccl:
            cnt = comm.rccl_ranks()
            if cnt != world:
                fails.append(f"rank {rank}: ncclCommCount says {cnt}, the launcher says WORLD_SIZE = {world} (a rank joined another communicator: check MASTER_PORT / the id hand-off)")
        # all-reduce(MAX), int32, odd length
        cnt = 1027
        i = np.arange(cnt, dtype=np.int64)
        mine = ((i * 7 + rank * 13) % 101 - 50).astype(np.int32)
        want = np.max(np.stack([((i * 7 + r * 13) % 101 - 50) for r in range(world)]), axis=0).astype(np.int32)
        buf = torch.from_numpy(mine).to(device)
        rc = c.allreduce_max_i32(c.ctx, buf.data_ptr(), cnt, st)
        sync()
        if rc != 0 or not np.array_equal(buf.cpu().numpy(), want):
            fails.append(f"rank {rank}: all-reduce(MAX, int32, {cnt}) rc = {rc}, {int((buf.cpu().numpy() != want).sum())} wrong entries")
        if world > 1:
            # grouped send / recv ring: to rank + 1, from rank - 1
            nb = 4096 + 8
            nxt, prv = (rank + 1) % world, (rank + world - 1) % world
            out = torch.from_numpy(((np.arange(nb) * 3 + rank * 17) % 251).astype(np.uint8)).to(device)
            inn = torch.zeros(nb, dtype=torch.uint8, device=device)
            ops = (P2POp * 2)(P2POp(out.data_ptr(), nb, nxt, 1), P2POp(inn.data_ptr(), nb, prv, 0))
            rc = c.sendrecv(c.ctx, 2, ops, st)
            sync()
            wantb = ((np.arange(nb) * 3 + prv * 17) % 251).astype(np.uint8)
            if rc != 0 or not np.array_equal(inn.cpu().numpy(), wantb):
                fails.append(f"rank {rank}: grouped send/recv ring ({nb} B to {nxt}, from {prv}) rc = {rc}, {int((inn.cpu().numpy() != wantb).sum())} wrong bytes")
        # reduce-scatter(sum), FP64: exact in double (small dyadic values)
        rcnt = 515
        w = np.arange(world, dtype=np.float64)[:, None]
        j = np.arange(rcnt, dtype=np.float64)[None, :]
        send = torch.from_numpy((rank + w * 0.5 + j * 0.25).reshape(-1)).to(device)
        recv = torch.zeros(rcnt, dtype=torch.float64, device=device)
        rc = c.reduce_scatter_sum_f64(c.ctx, send.data_ptr(), recv.data_ptr(), rcnt, st)
        sync()
        wantd = world * (world - 1) / 2.0 + world * (rank * 0.5 + np.arange(rcnt) * 0.25)
        if rc != 0 or not np.array_equal(recv.cpu().numpy(), wantd):
            fails.append(f"rank {rank}: reduce-scatter(sum, FP64, {rcnt} per rank) rc = {rc}, max abs deviation {float(np.abs(recv.cpu().numpy() - wantd).max())}")
    except Exception as e:  # a transport that throws is a failed self-test, with the reason
        return False, f"rank {rank}: transport self-test raised {type(e).__name__}: {e}"
    if fails:
        return False, "; ".join(fails)
    return True, "ok"


def _ld(t):
    """Leading dimension (elements between consecutive columns) of a column-major matrix held as a (cols, rows) tensor."""
    return t.stride(0) if t.shape[0] > 1 else max(t.shape[1], t.stride(0))


class DistGemm:
    """One sharded-GEMM plan (gemmul8_dist_create ... gemmul8_dist_destroy).  Matrices are column-major, passed as torch
    tensors of shape (cols, rows) like everywhere in this package, or as raw pointers + leading dimensions (run_ptr)."""

    def __init__(self, comm, plan, dtype_code, backend, m, n, k, N, fastmode=False, opA="N", opB="N", alpha=1.0, beta=0.0, engine=None,
                 grid_rows=0):
        import numpy as np
        self.lib = _lib()
        self.comm = comm
        self.kind = PLAN_CODES[plan] if isinstance(plan, str) else int(plan)
        self.m, self.n, self.k, self.N, self.fast = m, n, k, N, bool(fastmode)
        self.dt = dtype_code
        self.handle = C.c_void_p()
        self._engine = engine  # keep the callback table alive
        eng_ptr = C.byref(engine) if engine is not None else None
        g.check(self.lib.gemmul8_dist_create(comm.ptr, eng_ptr, self.kind, grid_rows, dtype_code, backend, g.OPS[opA], g.OPS[opB], m, n, k, N,
                                             int(fastmode), C.byref(self.handle)), "gemmul8_dist_create")
        np_dt = {g.S: np.float32, g.D: np.float64, g.Cx: np.complex64, g.Z: np.complex128}[dtype_code]
        self._alpha = np.array([alpha], dtype=np_dt)
        self._beta = np.array([beta], dtype=np_dt)
        mods, rows, cols = C.c_uint(0), C.c_size_t(0), C.c_size_t(0)
        g.check(self.lib.gemmul8_dist_my_work(self.handle, C.byref(mods), C.byref(rows), C.byref(cols)))
        self.my_planes, self.work_rows, self.work_cols = mods.value, rows.value, cols.value
        self.r0, self.r1, self.c0, self.c1 = self.owned_block(comm.rank)

    def owned_block(self, rank):
        v = [C.c_size_t(0) for _ in range(4)]
        g.check(self.lib.gemmul8_dist_owned_block(self.handle, rank, *[C.byref(x) for x in v]))
        return tuple(x.value for x in v)

    def workspace_bytes(self):
        return self.lib.gemmul8_dist_workspace_bytes(self.handle)

    def run_ptr(self, A, lda, B, ldb, Cp, ldc, stream=None):
        g.check(self.lib.gemmul8_dist_gemm(self.handle, stream, self._alpha.ctypes.data, A, lda, B, ldb, self._beta.ctypes.data, Cp, ldc),
                "gemmul8_dist_gemm")

    def run(self, A, B, Cmat, stream=None):
        if stream is None and Cmat.is_cuda:
            import torch
            stream = torch.cuda.current_stream(Cmat.device).cuda_stream
        self.run_ptr(A.data_ptr(), _ld(A), B.data_ptr(), _ld(B), Cmat.data_ptr(), _ld(Cmat), stream)

    def set_events(self, e0, e1):
        """torch.cuda.Event pair (already recorded once, so that the handles exist) recorded around this rank's low-precision GEMM."""
        g.check(self.lib.gemmul8_dist_set_events(self.handle, e0.cuda_event if e0 is not None else None, e1.cuda_event if e1 is not None else None))

    def set_exchange_events(self, events):
        """Four torch.cuda.Event (already recorded once) or None: [0], [1] around the all-reduce(MAX) of the bound maxima, [2], [3]
        around the bulk exchange (residue send/recv or FP64 reduce-scatter).  None clears."""
        if events is None:
            g.check(self.lib.gemmul8_dist_set_exchange_events(self.handle, None))
            return
        arr = (C.c_void_p * 4)(*[e.cuda_event if e is not None else None for e in events])
        g.check(self.lib.gemmul8_dist_set_exchange_events(self.handle, arr))

    def exchange_bytes(self):
        """(all-reduce payload, bytes sent to other ranks, bytes received from other ranks) per call on this rank."""
        v = [C.c_size_t(0) for _ in range(3)]
        g.check(self.lib.gemmul8_dist_exchange_bytes(self.handle, *[C.byref(x) for x in v]))
        return tuple(x.value for x in v)

    def gather_result(self, Cmat, stream=None):
        """Assemble the full C on every rank, in place."""
        if stream is None and Cmat.is_cuda:
            import torch
            stream = torch.cuda.current_stream(Cmat.device).cuda_stream
        g.check(self.lib.gemmul8_dist_allgather_c(self.handle, stream, Cmat.data_ptr(), _ld(Cmat)), "gemmul8_dist_allgather_c")
        return Cmat

    def close(self):
        if self.handle:
            self.lib.gemmul8_dist_destroy(self.handle)
            self.handle = C.c_void_p()

    def describe(self):
        if self.kind == BLOCKS:
            return (f"output blocks sharded over {self.comm.world} ranks (every rank runs all {self.N} moduli on its "
                    f"{self.work_rows} x {self.work_cols} block; one all_reduce(MAX) of int32[m+n] bounds over RCCL); A, B replicated on every rank")
        if self.kind == MODULI:
            return (f"moduli sharded x{self.comm.world} ({self.my_planes} of {self.N} moduli on this rank; grouped point-to-point exchange of INT8 "
                    f"residue blocks over RCCL, column-block CRT); A, B replicated on every rank")
        fg = os.environ.get("GEMMUL8_DIST_FP64_GROUPS", "")
        how = f"ncclReduceScatter(sum) per moduli group, {fg} groups, behind the next group's GEMMs" if fg.isdigit() and int(fg) > 1 and self.comm.world > 1 else "ncclReduceScatter(sum)"
        return (f"moduli sharded x{self.comm.world} ({self.my_planes} of {self.N} moduli on this rank; FP64 partial CRT sums, "
                f"{how}); A, B replicated on every rank")


def make_plan(comm, dtype_code, backend, m, n, k, N, mode=None, **kw):
    """A plan by name (blocks | columns | moduli | fp64sum); default: GEMMUL8_DIST_SHARD, else blocks."""
    mode = mode or os.environ.get("GEMMUL8_DIST_SHARD", "blocks")
    if mode == "columns":  # the 1 x G grid of the block plan
        kw["grid_rows"] = 1
        mode = "blocks"
    return DistGemm(comm, mode, dtype_code, backend, m, n, k, N, **kw)
