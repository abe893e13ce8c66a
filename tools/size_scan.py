import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import gemmul8_amd as g
for n in (1024, 2048, 4096, 8192):
    A = torch.rand((n, n), dtype=torch.float64, device="cuda") - 0.5
    B = torch.rand((n, n), dtype=torch.float64, device="cuda") - 0.5
    Cm = torch.zeros((n, n), dtype=torch.float64, device="cuda")
    tot, _, _ = g.work_size(False, g.INT8, n, n, n, 14); work = torch.empty(tot, dtype=torch.uint8, device="cuda")
    for fast in (False, True):
        for _ in range(5): g.gemm(A, B, 14, fastmode=fast, C_out=Cm, work=work)
        torch.cuda.synchronize(); reps = 200 if n <= 2048 else 30
        t0 = time.perf_counter()
        for _ in range(reps): g.gemm(A, B, 14, fastmode=fast, C_out=Cm, work=work)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
        print(f"n={n} fast={fast}: {dt*1e6:8.1f} us  {2*n**3/dt*1e-12:6.1f} TFLOPS")
    t0 = time.perf_counter()
    for _ in range(20): torch.mm(A, B, out=Cm)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print(f"n={n} native FP64: {2*n**3/dt*1e-12:6.1f} TFLOPS")
