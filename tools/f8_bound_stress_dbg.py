import sys, os
sys.path[:0] = [os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."), os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")]
import numpy as np
import gemmul8_amd as g, gpu_util as gu, oracle_lib as ol
import torch, ctypes as C
def case(seed):
    rng = np.random.default_rng(424242 + seed)
    cplx = seed % 3 == 2
    f32 = (seed // 3) % 2 == 1
    m = int(rng.choice([1, 2, 5, 17, 64])); n = int(rng.choice([1, 2, 3, 9, 48])); k = int(rng.choice([8, 64, 127, 400, 1024, 2300]))
    phi = float(rng.choice([2.0, 4.0, 6.0])); dens = float(rng.choice([0.1, 0.5, 1.0]))
    def mat(shape):
        x = (rng.random(shape) - 0.5) * np.exp(phi * rng.standard_normal(shape)) * (rng.random(shape) < dens)
        if cplx:
            x = x + 1j * (rng.random(shape) - 0.5) * np.exp(phi * rng.standard_normal(shape)) * (rng.random(shape) < dens)
        return x
    A, B = mat((m, k)), mat((k, n))
    A[0, 0] = A[0, 0] or 1.0; B[0, 0] = B[0, 0] or 1.0
    dt = (np.complex64 if f32 else np.complex128) if cplx else (np.float32 if f32 else np.float64)
    return A.astype(dt), B.astype(dt), (6 if f32 else 10), (m, n, k, phi, dens, cplx, f32)
for seed in map(int, sys.argv[1:]):
    A, B, N, info = case(seed)
    print(seed, info)
    dA, dB = gu.to_dev(A), gu.to_dev(B)
    m, k = A.shape; n = B.shape[1]
    tot, _, _ = g.work_size(False, g.FP8, m, n, k, N, 0, 0)
    work = torch.full((tot,), 0x5A, dtype=torch.uint8, device="cuda")
    L = g.Layout(); code = g._dtype_code(dA.dtype); lib = g.lib()
    g.check(lib.gemmul8_get_layout(code, g.FP8, m, n, k, N, work.data_ptr(), None, None, 0, 0, C.byref(L)))
    st = torch.cuda.current_stream().cuda_stream
    g.check(lib.gemmul8_scale_bounds(st, code, g.FP8, g.OPS["N"], g.OPS["N"], m, n, k, dA.data_ptr(), dA.shape[1], dB.data_ptr(), dB.shape[1], N, 0, n, C.byref(L), 0, 0))
    torch.cuda.synchronize()
    w = work.cpu().numpy(); base = work.data_ptr()
    np_ = (n + 255) // 256 * 256
    mx = w[L.scratch - base:L.scratch - base + 4 * (L.mp + np_)]
    rmax, cmax = mx[:4 * m].view(np.float32), mx[4 * L.mp:4 * L.mp + 4 * n].view(np.float32)
    oA, _ = ol.extract_bounds(A, "N", True, g.FP8); oB, _ = ol.extract_bounds(B, "N", False, g.FP8)
    orm, ocm = ol.bound_maxima(oA, oB, g.FP8)
    ex_r, ex_c = ol.bound_maxima_f8_exact(oA, oB)
    for d, o, ex, what in ((rmax, orm, ex_r, "row"), (cmax, ocm, ex_c, "col")):
        bad = np.nonzero(d.astype(np.float64) > o.astype(np.float64) * (1 + 2.0**-22))[0]
        for i in bad[:4]:
            print(" ", what, i, "dev", float(d[i]).hex(), "oracle", float(o[i]).hex(), "exact", float(ex[i]).hex(), "dev/or-1", d[i] / o[i] - 1, "kp", L.kp)
