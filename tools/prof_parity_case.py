import sys, os, time, cProfile, pstats
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np, torch
import gemmul8_amd as g, gpu_util as gu
rng = np.random.default_rng(0)
A = (rng.random((37, 300)) - 0.5); B = (rng.random((300, 41)) - 0.5)
gu.parity_case(A, B, 14, False)
t0 = time.time()
for _ in range(5): gu.parity_case(A, B, 14, False)
print("parity_case accurate d N=14:", (time.time() - t0) / 5)
pr = cProfile.Profile(); pr.enable()
for _ in range(3): gu.parity_case(A, B, 14, False)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
