"""CPU (gloo, world sizes 2..8) tests of the multi-GPU plans THROUGH THE C ABI of include/gemmul8_dist.h: the C++ code of
gemmul8_amd/csrc/oz2_dist.cpp (partition arithmetic, bound all-reduce, residue exchange, FP64 partial-sum reduce-scatter,
allgather of C) runs for real; only the per-rank compute and memory functions are replaced by tests/dist_cpu_engine.py (the
CPU oracle on host memory in the device's workspace layout) and the transport by gloo (gemmul8_amd.dist.TorchTransport).
The assembled result must be bit-identical to the single-process oracle for the block and moduli plans at every world size;
the FP64-sum plan must equal the oracle's rank-grouped accumulation (exact at world 2, where the summation order is unique)."""
import ctypes as C
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib as ol

NP_DT = {"d": np.float64, "s": np.float32, "z": np.complex128}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rand(shape, dtype, rng):
    x = (rng.random(shape) - 0.5) * np.exp(rng.standard_normal(shape))
    if np.dtype(dtype).kind == "c":
        x = x + 1j * (rng.random(shape) - 0.5) * np.exp(rng.standard_normal(shape))
    return np.asfortranarray(x.astype(dtype))


def split_range(total, parts, idx):
    q, r = divmod(total, parts)
    b = idx * q + min(idx, r)
    return b, b + q + (1 if idx < r else 0)


def _worker(rank, world, port, plan, N, fast, m, n, k, typ, opA, opB, grid_rows, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import gemmul8_amd as g
        from gemmul8_amd import dist as gd
        from dist_cpu_engine import OracleEngine
        dtype = NP_DT[typ]
        code = ol.DT[np.dtype(dtype)]
        rng = np.random.default_rng(42)
        A = _rand((m, k) if opA == "N" else (k, m), dtype, rng)
        B = _rand((k, n) if opB == "N" else (n, k), dtype, rng)
        C0 = _rand((m, n), dtype, rng)
        alpha, beta = (-1.5, 1.5) if np.dtype(dtype).kind != "c" else (-1.5 + 0.5j, 1.5 - 0.25j)
        eng = OracleEngine()
        comm = gd.TorchTransport(device=False)
        pl = gd.DistGemm(comm, plan, code, g.INT8, m, n, k, N, fastmode=fast, opA=opA, opB=opB, alpha=alpha, beta=beta, engine=eng.table,
                         grid_rows=grid_rows)
        ldc = m + 3  # a C with padding between columns
        Cbuf = np.full((n, ldc), 7.25, dtype)          # row j = column j of the column-major matrix
        Cbuf[:, :m] = C0.T
        pl.run_ptr(A.ctypes.data, A.shape[0], B.ctypes.data, B.shape[0], Cbuf.ctypes.data, ldc)
        r0, r1, c0, c1 = pl.owned_block(rank)
        untouched = Cbuf.copy()
        g.check(pl.lib.gemmul8_dist_allgather_c(pl.handle, None, Cbuf.ctypes.data, ldc))
        assert np.array_equal(Cbuf[c0:c1, r0:r1].view(np.uint8), untouched[c0:c1, r0:r1].view(np.uint8)), "allgather changed the own block"
        assert np.all(Cbuf[:, m:] == 7.25), "padding between the columns of C was written"
        calls = eng.calls
        # accounting of gemmul8_dist_exchange_bytes: what the ranks send in total is what they receive in total; the block plan moves
        # nothing but the bounds vector; exchange events are accepted (and ignored) with a non-HIP engine
        ar, tx, rx = pl.exchange_bytes()
        pl.set_exchange_events(None)
        import torch
        tot = torch.tensor([tx, rx], dtype=torch.int64)
        dist.all_reduce(tot)
        assert int(tot[0]) == int(tot[1]), (plan, tot)
        if plan == "blocks":
            assert tx == 0 and rx == 0 and ar == (0 if fast or world == 1 else 4 * (m + n))
        else:
            assert (ar > 0) == (not fast and world > 1)
        assert comm.rccl_ranks() == -1
        pl.close()
        if rank == 0:
            full = np.ascontiguousarray(Cbuf[:, :m].T)
            if plan == "fp64sum":
                _, it = ol.gemm(A, B, N, fastmode=fast, opA=opA, opB=opB, want_intermediates=True)
                ref = np.asfortranarray(C0).copy(order="F")
                bounds = np.array([split_range(N, world, r)[0] for r in range(world)] + [N], np.uint32)
                al, be = np.array([alpha], dtype), np.array([beta], dtype)
                ol.lib().oz2_invscal_grouped(code, 0, N, m, n, ol._p(it["C_mid"]), ol._p(it["sftA"]), ol._p(it["sftB"]), ol._p(al), ol._p(be),
                                             ol._p(ref), m, 0, world, ol._p(bounds))
                single = ol.gemm(A, B, N, fastmode=fast, opA=opA, opB=opB, alpha=alpha, beta=beta, C0=C0)
                q.put(("fp64sum", full.tobytes() == np.ascontiguousarray(ref).tobytes(), int((full != single).sum()), full.size,
                       float(np.max(np.abs(full - single) / np.maximum(np.abs(single), 1e-300))), sum(c[0] == "add_f64" for c in calls)))
            else:
                ref = ol.gemm(A, B, N, fastmode=fast, opA=opA, opB=opB, alpha=alpha, beta=beta, C0=C0)
                q.put((plan, full.tobytes() == np.ascontiguousarray(ref).tobytes(), calls))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _start_and_reap(procs, timeout):
    """Start the ranks, wait, and KILL whatever is still alive afterwards: a rank that died (port taken, runtime error) leaves its peer
    inside a collective, and multiprocessing joins non-daemon children at interpreter exit -- one flaky rendezvous then blocks the
    whole pytest process until the collective's own timeout (half an hour)."""
    for p in procs:
        p.daemon = True
        p.start()
    import time
    deadline = time.time() + timeout
    for p in procs:
        p.join(max(0.0, deadline - time.time()))
        if p.exitcode not in (0, None):   # a rank failed: its peers will never finish
            break
    codes = [p.exitcode for p in procs]
    for p in procs:
        if p.is_alive():
            p.kill()
            p.join(10)
    return codes


def _run(world, plan, N, fast, m, n, k, typ="d", opA="N", opB="N", grid_rows=0):
    ctx = mp.get_context("spawn")
    for attempt in range(2):   # the free port is chosen before the ranks bind it: one retry for the rare rendezvous collision
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, world, port, plan, N, fast, m, n, k, typ, opA, opB, grid_rows, q)) for r in range(world)]
        codes = _start_and_reap(procs, 300)
        if codes == [0] * len(procs):
            return q.get(timeout=10)
    raise AssertionError(f"ranks exited with {codes}")


@pytest.mark.parametrize("world,grid_rows", [(2, 0), (4, 0), (4, 4), (3, 0), (8, 0), (2, 1)])
@pytest.mark.parametrize("N,fast,m,n", [(14, False, 19, 11), (9, True, 19, 11), (5, False, 3, 2)])
def test_block_plan_bit_identical(world, grid_rows, N, fast, m, n):
    """Output blocks on a Gr x Gc rank grid (default 2 -> 2x1, 4 -> 2x2, 8 -> 4x2; grid_rows = 1 is the column plan): one
    all_reduce(MAX) of the int32[m + n] bound vector; m = 3, n = 2 leaves ranks without a block that still join the collective."""
    _, same, calls = _run(world, "blocks", N, fast, m, n, 37)
    assert same, "block-sharded result differs from the single-process oracle"


@pytest.mark.parametrize("opA,opB,typ", [("T", "N", "d"), ("N", "T", "d"), ("C", "T", "z"), ("N", "N", "s")])
def test_block_plan_ops_and_types(opA, opB, typ):
    """Row blocks of op(A) / column blocks of op(B) are strided views that depend on the op: all of them, complex and float too."""
    _, same, _ = _run(4, "blocks", 12 if typ == "s" else 15, False, 21, 13, 40, typ=typ, opA=opA, opB=opB)
    assert same


@pytest.mark.parametrize("world", [2, 3, 4, 8])
@pytest.mark.parametrize("N,fast,n", [(14, False, 11), (9, True, 11), (2, False, 5), (14, True, 2)])
def test_moduli_plan_bit_identical(world, N, fast, n):
    """Moduli sharded, INT8 residue exchange: ranks without moduli (N = 2 on 3+ ranks) or without columns (n = 2) still take part;
    the engine poisons every plane outside the rank's moduli, so a result equal to the oracle proves they are never read."""
    _, same, calls = _run(world, "moduli", N, fast, 19, n, 37)
    assert same, "moduli-sharded result differs from the single-process oracle"
    # rank 0 multiplies exactly its share, in ascending groups (round 5: the exchange of a group travels while the next group multiplies)
    lp = [(c[3], c[4]) for c in calls if c[0] == "lowprec"]
    t0, t1 = split_range(N, world, 0)
    assert lp and lp[0][0] == t0 and lp[-1][1] == t1 and all(a[1] == b[0] for a, b in zip(lp, lp[1:])), lp


@pytest.mark.parametrize("groups", [1, 2, 3])
def test_moduli_plan_exchange_groups(groups, monkeypatch):
    """GEMMUL8_DIST_GROUPS: the rank's planes go out in 1 (the round-4 order), 2 (default) or 3 groups, each followed by its grouped send / recv;
    every rank cuts every sender's planes the same way (world 4, N = 14: 4 + 4 + 3 + 3 planes, so the groups differ from rank to rank) and the
    result stays bit-identical."""
    monkeypatch.setenv("GEMMUL8_DIST_GROUPS", str(groups))
    _, same, calls = _run(4, "moduli", 14, False, 19, 11, 37)
    assert same
    lp = [(c[3], c[4]) for c in calls if c[0] == "lowprec"]
    assert len(lp) == groups and lp[0][0] == 0 and lp[-1][1] == 4, lp


def test_moduli_plan_complex_transposed():
    _, same, _ = _run(3, "moduli", 13, False, 17, 9, 33, typ="z", opA="C", opB="T")
    assert same


@pytest.mark.parametrize("N,fast,typ", [(14, False, "d"), (9, True, "d"), (12, False, "s"), (15, False, "z")])
def test_fp64sum_plan_world2_matches_grouped_oracle(N, fast, typ):
    """Exchange variant (A): FP64 partial CRT sums + reduce-scatter(sum).  With two ranks the sum of the partials has one
    possible order, so the result must equal the oracle's rank-grouped accumulation bit for bit; against the single-GPU result it
    may differ in the last bits of a few elements (the rounded lo chain -- or, for float outputs, the single chain -- is grouped
    differently): counted and bounded here, measured at full size on the GPU (tests/test_gpu_dist.py)."""
    _, same, nbad, total, rel, nadd = _run(2, "fp64sum", N, fast, 19, 11, 37, typ=typ)
    assert nadd == 0
    assert same, "FP64-sum plan differs from the oracle's grouped accumulation"
    eps = 2.0 ** -22 if typ == "s" else 2.0 ** -50
    assert rel <= eps, (nbad, total, rel)


def test_fp64sum_plan_world4_close_to_single():
    """Four ranks: the transport's summation order is not specified; integer intermediates are identical, the final values agree
    with the single-process result to the last bits."""
    _, same, nbad, total, rel, _ = _run(4, "fp64sum", 15, False, 19, 11, 37)
    assert rel <= 2.0 ** -50, (nbad, total, rel)


def test_rccl_symbols_are_optional_at_load_time():
    """libgemmul8.so must load (and every non-dist entry point work) on a machine without librccl: RCCL is bound at run time."""
    import subprocess
    import gemmul8_amd as g
    out = subprocess.run(["readelf", "-d", g.LIB_PATH], capture_output=True, text=True).stdout
    assert "rccl" not in out and "amdhip64" not in out and "hipblas" not in out, out


def _id_worker(rank, world, port, q):
    import ctypes as C
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", GEMMUL8_DIST_PORT=str(port))
    from gemmul8_amd import dist as gd
    buf = C.create_string_buffer(128)
    r, w = C.c_int(-1), C.c_int(-1)
    rc = gd._lib().gemmul8_comm_rccl_id_from_env(buf, C.byref(r), C.byref(w))
    q.put((rank, rc, r.value, w.value, buf.raw))


def test_unique_id_rendezvous_over_tcp():
    """gemmul8_comm_rccl_id_from_env: rank 0 creates the ncclUniqueId and hands it to the other ranks over TCP (the bootstrap a C++ host
    or the hook's GEMMUL8_DIST uses when there is no torch.distributed): every rank ends up with the same 128 bytes.  (Creating the
    communicator itself needs GPUs: tests/test_gpu_dist.py.)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    world = 4
    procs = [ctx.Process(target=_id_worker, args=(r, world, port, q)) for r in range(world)]
    for p in reversed(procs):   # rank 0 last: the others must retry until it listens
        p.start()
    for p in procs:
        p.join(120)
    codes = [p.exitcode for p in procs]
    for p in procs:
        if p.is_alive():
            p.kill()
    assert codes == [0] * len(procs), codes
    got = sorted(q.get(timeout=10) for _ in range(world))
    assert [g_[0] for g_ in got] == list(range(world))
    assert all(g_[1] == 0 and g_[2] == g_[0] and g_[3] == world for g_ in got), [(g_[0], g_[1]) for g_ in got]
    assert len({g_[4] for g_ in got}) == 1 and any(got[0][4]), "ranks disagree on the unique id"


def _selftest_worker(rank, world, port, break_it, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gemmul8_amd import dist as gd
        comm = gd.TorchTransport(device=False)
        if break_it and rank == 1:
            # a transport whose reduce-scatter returns garbage on one rank: the self-test must say so, with the operation's name
            def bad(ctx, send, recv, cnt, stream):
                rc = comm._redscat(ctx, send, recv, cnt, stream)
                (C.c_double * 1).from_address(recv)[0] += 1.0
                return rc
            comm._keep.append(gd.REDSCAT_FN(bad))
            comm.struct.reduce_scatter_sum_f64 = comm._keep[-1]
        ok, msg = gd.selftest(comm, "cpu", None, expect_rccl=False)
        q.put((rank, ok, msg))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("break_it", [False, True])
def test_transport_selftest_over_gloo(world, break_it):
    """gemmul8_amd.dist.selftest (what `bench.py --gpus N` runs before it times anything): all-reduce(MAX, int32), grouped send/recv ring and
    reduce-scatter(sum, FP64) checked against host arithmetic on every rank; a transport that delivers one wrong value is reported with
    the name of the operation and the rank."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_selftest_worker, args=(r, world, port, break_it, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    for rank, ok, msg in res:
        if break_it and rank == 1:
            assert not ok and "reduce-scatter" in msg and "rank 1" in msg, msg
        else:
            assert ok and msg == "ok", (rank, msg)


def _id_worker_env(rank, world, port, env, q):
    import ctypes as C
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), GEMMUL8_DIST_PORT=str(port), MASTER_PORT=str(port - 17))
    os.environ.update(env)
    from gemmul8_amd import dist as gd
    buf = C.create_string_buffer(128)
    r, w = C.c_int(-1), C.c_int(-1)
    rc = gd._lib().gemmul8_comm_rccl_id_from_env(buf, C.byref(r), C.byref(w))
    q.put((rank, rc, buf.raw))


def _run_id_rendezvous(envs, timeout=60):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_id_worker_env, args=(r, len(envs), port, envs[r], q)) for r in range(len(envs))]
    _start_and_reap(procs, timeout)
    return sorted(q.get(timeout=10) for _ in range(len(envs)))


def test_unique_id_rendezvous_when_master_addr_is_the_debian_hostname_alias():
    """on rank 0, MASTER_ADDR may resolve to 127.0.1.1 (Debian / Ubuntu map the host's own name there); every 127/8
    address is locally bindable, so binding the resolved address 'works' and listens on loopback only -- ranks on other nodes never get
    through.  Rank 0 must listen on all interfaces in that case: a peer that reaches the host through a DIFFERENT address (here 127.0.0.1
    stands in for the NIC address) gets the id."""
    got = _run_id_rendezvous([{"MASTER_ADDR": "127.0.1.1"}, {"MASTER_ADDR": "127.0.0.1"}])
    assert all(rc == 0 for _, rc, _ in got), got
    assert got[0][2] == got[1][2] and any(got[0][2])


def test_unique_id_rendezvous_refuses_a_rank_with_the_wrong_job_secret():
    """Hand-off protocol 2 (csrc/oz2_dist.cpp): rank 0 sends a fresh salt, the client answers SipHash-2-4(key(GEMMUL8_DIST_SECRET); salt | rank |
    world | port).  A rank holding another secret is never served: both sides end with an error after GEMMUL8_DIST_TIMEOUT instead of a
    communicator id in the wrong hands (or a hang)."""
    base = {"MASTER_ADDR": "127.0.0.1", "GEMMUL8_DIST_TIMEOUT": "3"}
    got = _run_id_rendezvous([dict(base, GEMMUL8_DIST_SECRET="job-4711"), dict(base, GEMMUL8_DIST_SECRET="job-4712")])
    assert all(rc != 0 for _, rc, _ in got), got
    good = _run_id_rendezvous([dict(base, GEMMUL8_DIST_SECRET="job-4711"), dict(base, GEMMUL8_DIST_SECRET="job-4711")])
    assert all(rc == 0 for _, rc, _ in good) and good[0][2] == good[1][2]


def _groups_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["GEMMUL8_DIST_GROUPS"] = "2" if rank == 0 else "3"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import gemmul8_amd as g
        from gemmul8_amd import dist as gd
        from dist_cpu_engine import OracleEngine
        rng = np.random.default_rng(1)
        m, n, k, N = 12, 10, 16, 8
        A, B = _rand((m, k), np.float64, rng), _rand((k, n), np.float64, rng)
        Cbuf = np.zeros((n, m))
        eng = OracleEngine()
        comm = gd.TorchTransport(device=False)
        pl = gd.DistGemm(comm, "moduli", ol.DT[np.dtype(np.float64)], g.INT8, m, n, k, N, fastmode=True, engine=eng.table)
        try:
            pl.run_ptr(A.ctypes.data, m, B.ctypes.data, k, Cbuf.ctypes.data, m)
            q.put((rank, "ran"))
        except RuntimeError as e:
            q.put((rank, str(e)))
        pl.close()
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_moduli_plan_refuses_ranks_that_disagree_on_the_group_count():
    """GEMMUL8_DIST_GROUPS is read per rank; a mismatch would give unequal grouped send / recv counts per pair -- a silent
    hang with RCCL.  The first call cross-checks it with one small all-reduce and fails on EVERY rank with GEMMUL8_E_ARG."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_groups_worker, args=(r, 2, port, q)) for r in range(2)]
    codes = _start_and_reap(procs, 120)
    got = sorted(q.get(timeout=10) for _ in range(2))
    assert codes == [0, 0], codes
    assert all("failed with status" in msg for _, msg in got), got


@pytest.mark.parametrize("world,groups,N,typ", [(2, 2, 14, "d"), (2, 3, 15, "z"), (4, 2, 15, "d"), (3, 8, 13, "d"), (8, 2, 14, "d"), (2, 2, 12, "s")])
def test_fp64sum_plan_in_moduli_groups(world, groups, N, typ, monkeypatch):
    """GEMMUL8_DIST_FP64_GROUPS (round 6): the rank's planes are multiplied group by group, each group's FP64 partial sums go through their own
    reduce-scatter (beside the next group's GEMMs on the HIP engine) and the reduced blocks are added up.  Integer intermediates are those of the
    single collective; the hi parts add exactly in any order, the lo parts see one more rounding per group: the final values agree with the
    single-process result to the last bits (same bound as the one-collective plan), with ranks that own fewer planes than there are groups
    (world 8, N = 14: 2 or 1 planes; world 3 with 8 groups) contributing zeros in their empty groups."""
    monkeypatch.setenv("GEMMUL8_DIST_FP64_GROUPS", str(groups))
    _, same, nbad, total, rel, nadd = _run(world, "fp64sum", N, False, 19, 11, 37, typ=typ)
    assert nadd == groups - 1, nadd     # the grouped schedule really ran: one running-sum pass per group after the first
    eps = 2.0 ** -22 if typ == "s" else 2.0 ** -50
    assert rel <= eps, (nbad, total, rel)
    assert nbad <= max(1, 0.01 * total), (nbad, total)
