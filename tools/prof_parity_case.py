"""cProfile of one sub-matrix-view parity test body (tests/test_gpu_ld.py): where the GPU suite's per-test second goes."""
import sys, os, time, cProfile, pstats
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
import gemmul8_amd as g, gpu_util as gu
def body(variant="fp6", dtype="complex128", N=12):
    rng = np.random.default_rng(0)
    dt = np.dtype(dtype)
    def rand(shape):
        x = (rng.random(shape) - 0.5)
        if dt.kind == "c": x = x + 1j * (rng.random(shape) - 0.5)
        return x.astype(dt)
    A, B, C0 = rand((72, 47)), rand((47, 67)), rand((72, 67))
    for i in range(3):
        gu.parity_case_embedded(A, B, C0, N, bool(i & 1), "N", "N", 0.75, -0.5, g.FP8 if variant != "int8" else g.INT8, (1, 7, 64), (1, 3, 1), rng)
body(); body("int8", "float64", 14)
for args in (("fp6", "complex128", 12), ("int8", "float64", 14), ("fp6", "float32", 6)):
    t0 = time.time(); body(*args); print(args, "body:", time.time() - t0)
pr = cProfile.Profile(); pr.enable(); body(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
