"""Eager launches vs HIP-graph replay of gemmul8_gemm for launch-bound shapes (DGEMM, 14 moduli, accurate mode)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gemmul8_amd as g  # noqa: E402

for n in (256, 512, 1024, 2048, 4096):
    A = torch.rand((n, n), dtype=torch.float64, device="cuda") - 0.5
    B = torch.rand((n, n), dtype=torch.float64, device="cuda") - 0.5
    Cm = torch.zeros((n, n), dtype=torch.float64, device="cuda")
    tot, _, _ = g.work_size(False, g.INT8, n, n, n, 14)
    work = torch.empty(tot, dtype=torch.uint8, device="cuda")
    for _ in range(3):
        g.gemm(A, B, 14, C_out=Cm, work=work)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        g.gemm(A, B, 14, C_out=Cm, work=work)
    reps = 200 if n <= 1024 else 50

    def timeit(fn):
        fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / reps
    te = timeit(lambda: g.gemm(A, B, 14, C_out=Cm, work=work))
    tg = timeit(graph.replay)
    tn = timeit(lambda: torch.matmul(B, A))
    f = 2.0 * n ** 3 * 1e-12
    print(f"n={n:5d}  eager {te*1e6:8.1f} us ({f/te:6.1f} TFLOPS)   graph {tg*1e6:8.1f} us ({f/tg:6.1f} TFLOPS)   native fp64 {tn*1e6:8.1f} us ({f/tn:6.1f} TFLOPS)")
