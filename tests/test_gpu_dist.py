"""-m gpu: the moduli-sharded driver with the REAL HipEngine, two ranks sharing cuda:0 (gloo, host-staged
exchange -- the GPU box has one GPU; the 8-GPU run uses the same code with NCCL/RCCL).  The assembled
result must be bit-identical to the single-GPU C-ABI result."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, N, fast, m, n, k, q, exchange="p2p"):
    os.environ["GEMMUL8_DIST_EXCHANGE"] = exchange if exchange in ("p2p", "a2a") else "p2p"
    os.environ["GEMMUL8_DIST_SHARD"] = {"columns": "columns", "blocks": "blocks", "blocks_rows": "blocks"}.get(exchange, "moduli")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import gemmul8_amd as g
        from gemmul8_amd import dist as gd
        torch.cuda.set_device(0)
        rng = np.random.default_rng(4)
        A = torch.from_numpy(rng.random((k, m)) - 0.5).cuda()  # (cols, rows) = column-major m x k
        B = torch.from_numpy(rng.random((n, k)) - 0.5).cuda()
        Cm = torch.zeros((n, m), dtype=torch.float64, device="cuda")
        if exchange == "blocks_rows":   # 2 x 1 grid: strided row-block views of A and C
            plan = gd.BlockShardedGemm(g.D, g.INT8, m, n, k, N, fastmode=fast, device=torch.device("cuda", 0), grid=(2, 1))
        else:
            plan = gd.make_plan(g.D, g.INT8, m, n, k, N, fastmode=fast, device=torch.device("cuda", 0))
        plan.run(A, B, Cm)
        torch.cuda.synchronize()
        full = plan.gather_result(Cm)
        if rank == 0:
            ref, _, _ = g.gemm(A, B, N, fastmode=fast)
            torch.cuda.synchronize()
            q.put(bool(torch.equal(full, ref)))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("exchange", ["p2p", "a2a", "columns", "blocks", "blocks_rows"])
@pytest.mark.parametrize("N,fast", [(14, False), (15, True)])
def test_two_ranks_one_gpu_bitwise(N, fast, exchange):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    m, n, k = 300, 515, 1000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, N, fast, m, n, k, q, exchange)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert q.get(timeout=10)


@pytest.mark.parametrize("shard,ranks", [("blocks", 2), ("moduli", 2), ("blocks", 8)])
def test_bench_multi_rank_contract(shard, ranks):
    """bench.py as the driver launches it for N > 1 (torch.distributed.run, one rank per process), here with 2 or 8 ranks sharing
    the box's single GPU over gloo (GEMMUL8_DIST_BACKEND: NCCL refuses duplicate devices): one JSON line from rank 0 with the
    contract keys, the right rank count and an accurate result."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, GEMMUL8_DIST_BACKEND="gloo", GEMMUL8_DIST_SHARD=shard)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", str(ranks), "--steps", "2", "--warmup", "1",
                          "--size", "1024"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline"):
        assert key in d, key
    assert d["n_gpus"] == ranks and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "strong" and d["value"] > 0
    assert d["max_rel_err"] < 1e-9


def _rccl_worker(port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        mx = torch.arange(-5, 1019, dtype=torch.int32, device=dev)
        ref = mx.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)                      # the bound exchange of the block / column plans
        blk = torch.rand((64, 48), dtype=torch.float64, device=dev)
        dist.broadcast(blk, src=0)                                     # gather_result
        send = torch.randint(0, 256, (4096,), dtype=torch.uint8, device=dev)
        recv = torch.empty_like(send)
        dist.all_to_all_single(recv, send)                             # residue exchange of the moduli plan (a2a)
        t = torch.tensor([1.5], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)                       # bench.py's max-over-ranks timing
        dist.barrier()
        torch.cuda.synchronize()
        q.put(bool(torch.equal(mx, ref)) and bool(torch.equal(recv, send)) and float(t.item()) == 1.5)
    finally:
        dist.destroy_process_group()


def test_rccl_collectives_used_by_the_plans_world1():
    """The RCCL calls the multi-GPU plans and bench.py make (int32 all_reduce(MAX), broadcast, all_to_all_single, barrier) on the real
    backend with the one GPU this box has: proves the library, the device_id init and the dtype/op combinations work here."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    p = ctx.Process(target=_rccl_worker, args=(port, q))
    p.start()
    p.join(300)
    assert p.exitcode == 0
    assert q.get(timeout=10)


@pytest.mark.parametrize("shard", ["blocks", "columns", "moduli"])
def test_bench_plan_path_on_real_rccl_one_rank(shard):
    """bench.py's multi-GPU code path (nccl process group with device_id, plan, barrier, max-over-ranks timing, gather) on the REAL RCCL
    backend with the single rank this box allows (GEMMUL8_BENCH_FORCE_PLAN=1)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, GEMMUL8_BENCH_FORCE_PLAN="1", GEMMUL8_DIST_SHARD=shard)
    env.pop("GEMMUL8_DIST_BACKEND", None)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                          "--size", "2048"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 10 and d["max_rel_err"] < 1e-9 and d["roofline"]["achieved"] > 0
