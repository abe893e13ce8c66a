import ctypes as C, os, sys, shutil, tempfile, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch, numpy as np
import gemmul8_amd as g
ref = g.lib()
n, N = 8192, 20
libs = []
tmp = tempfile.mkdtemp()
for i, pth in enumerate(sys.argv[1:]):
    cp = os.path.join(tmp, f"v{i}.so"); shutil.copy(pth, cp)
    L = C.CDLL(cp); L.gemmul8_lowprec_gemm.restype = C.c_int; L.gemmul8_lowprec_gemm.argtypes = ref.gemmul8_lowprec_gemm.argtypes
    libs.append(L)
tot, _, _ = g.work_size(True, g.INT8, n, n, n, N)
work = torch.empty(tot, dtype=torch.uint8, device="cuda")
Lo = g.Layout(); g.check(ref.gemmul8_get_layout(g.Z, g.INT8, n, n, n, N, work.data_ptr(), None, None, 0, 0, C.byref(Lo)))
work.random_(0, 256)
st = torch.cuda.current_stream().cuda_stream
ts = [[] for _ in libs]
for r in range(6):
    for i, L in enumerate(libs):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.check(L.gemmul8_lowprec_gemm(st, g.Z, g.INT8, n, n, n, N, 0, N, C.byref(Lo))); e1.record(); torch.cuda.synchronize()
        if r >= 1: ts[i].append(e0.elapsed_time(e1))
for i, pth in enumerate(sys.argv[1:]):
    t = sorted(ts[i]); print(f"{os.path.basename(pth):24s} lowprec phase (ZGEMM 8192^3, 20 moduli) median {t[len(t)//2]:7.2f} ms  min {t[0]:7.2f}")
