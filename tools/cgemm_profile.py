import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import gemmul8_amd as g
n = 8192; N = 7
A = torch.complex(torch.rand((n, n), dtype=torch.float32, device="cuda") - 0.5, torch.rand((n, n), dtype=torch.float32, device="cuda") - 0.5)
B = torch.complex(torch.rand((n, n), dtype=torch.float32, device="cuda") - 0.5, torch.rand((n, n), dtype=torch.float32, device="cuda") - 0.5)
Cm = torch.zeros((n, n), dtype=torch.complex64, device="cuda")
tot, _, _ = g.work_size(True, g.INT8, n, n, n, N); work = torch.empty(tot, dtype=torch.uint8, device="cuda")
for fast in (False, True):
    for _ in range(12): g.gemm(A, B, N, fastmode=fast, C_out=Cm, work=work)
torch.cuda.synchronize()
