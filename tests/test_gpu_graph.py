"""The whole emulated GEMM (memset, scale, bound GEMM, shifts, quantise, batched GEMMs, CRT) is a fixed sequence of
asynchronous launches on the caller's stream with no host synchronisation, so it can be captured in a HIP graph and replayed
(launch-bound small shapes).  Replays must reproduce the eager result bit for bit, also after the inputs changed in place."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dt,backend,N,fast", [(torch.float64, "INT8", 14, False), (torch.float64, "INT8", 14, True),
                                               (torch.float32, "FP8", 6, False), (torch.float32, "INT8", 7, False),
                                               (torch.complex64, "FP8", 6, False), (torch.complex128, "INT8", 20, False)])
def test_graph_capture_replay_bit_exact(dt, backend, N, fast):
    import gemmul8_amd as g
    be = getattr(g, backend)
    m, n, k = 520, 392, 1031
    gen = torch.Generator(device="cuda").manual_seed(3)
    rdt = torch.float32 if dt in (torch.float32, torch.complex64) else torch.float64

    def rnd(shape):
        x = torch.rand(shape, generator=gen, dtype=rdt, device="cuda") - 0.5
        if dt.is_complex:
            x = torch.complex(x, torch.rand(shape, generator=gen, dtype=rdt, device="cuda") - 0.5)
        return x.contiguous()
    A, B = rnd((k, m)), rnd((n, k))
    Cg = torch.zeros((n, m), dtype=dt, device="cuda")
    tot, _, _ = g.work_size(dt.is_complex, be, m, n, k, N)
    work = torch.empty(tot, dtype=torch.uint8, device="cuda")
    g.gemm(A, B, N, fastmode=fast, backend=be, C_out=Cg, work=work)        # warm-up outside capture (function attributes)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        g.gemm(A, B, N, fastmode=fast, backend=be, C_out=Cg, work=work)
    for trial in range(6):
        A.copy_(rnd((k, m)) * (2.0 ** (trial % 3)))
        B.copy_(rnd((n, k)))
        Cg.zero_()
        graph.replay()
        torch.cuda.synchronize()
        Ce, _, _ = g.gemm(A, B, N, fastmode=fast, backend=be)
        torch.cuda.synchronize()
        assert torch.equal(Cg, Ce), f"replay {trial} differs from the eager call"
