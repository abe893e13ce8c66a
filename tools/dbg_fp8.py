import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np, gemmul8_amd as g, gpu_util as gu, oracle_lib as ol
rng = np.random.default_rng(600)
m, n, k = 5, 4, 12
A = (rng.random((m, k)) - 0.5); B = (rng.random((k, n)) - 0.5)
for N, fast in ((2, True), (7, True)):
    Cd, it = gu.hip_gemm(A, B, N, fastmode=fast, backend=g.FP8, want_intermediates=True)
    Co, ito = ol.gemm(A, B, N, fastmode=fast, backend=g.FP8, sftA_in=it["sftA"], sftB_in=it["sftB"], want_intermediates=True)
    print("N", N, "sftA", it["sftA"], ito["sftA"])
    for q in range(it["A_lo"].shape[1]):
        d = it["A_lo"][0, q]; o = ito["A_lo"][0, q]
        print(" plane", q, "equal" if np.array_equal(d, o) else "DIFF")
        if not np.array_equal(d, o):
            idx = np.argwhere(d != o)[:6]
            for i, kk in idx: print("   ", i, kk, "dev", hex(d[i, kk]), "orc", hex(o[i, kk]))
    print(" C_mid equal", np.array_equal(it["C_mid"], ito["C_mid"]), " C equal", np.array_equal(Cd, Co))
