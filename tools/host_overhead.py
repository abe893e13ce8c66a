import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, numpy as np, gemmul8_amd as g
for n in (256, 512, 2048):
    gen = torch.Generator(device="cuda").manual_seed(1)
    A = torch.rand((n, n), generator=gen, dtype=torch.float64, device="cuda") - 0.5
    B = torch.rand((n, n), generator=gen, dtype=torch.float64, device="cuda") - 0.5
    Cm = torch.zeros((n, n), dtype=torch.float64, device="cuda")
    tot, _, _ = g.work_size(False, g.INT8, n, n, n, 14)
    work = torch.empty(tot, dtype=torch.uint8, device="cuda")
    for _ in range(5): g.gemm(A, B, 14, C_out=Cm, work=work)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200): g.gemm(A, B, 14, C_out=Cm, work=work)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 200
    print(f"n={n} ASYNC={os.environ.get('GEMMUL8_ASYNC','0')}: {dt*1e6:.1f} us per call (wall, 200 back-to-back calls)")
