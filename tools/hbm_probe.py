"""Achievable HBM streaming rates on this board (torch copy / fill / reduce over 2 GiB), the yardstick for the scale and CRT kernels."""
import time
import torch

n = 1 << 28  # 2 GiB of float64
x = torch.rand(n, dtype=torch.float64, device="cuda")
y = torch.empty_like(x)


def t(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


c = t(lambda: y.copy_(x))
f = t(lambda: y.fill_(1.0))
r = t(lambda: x.sum())
m = t(lambda: torch.amax(x))
print(f"copy  {2 * n * 8 / c * 1e-12:.2f} TB/s (read + write)")
print(f"fill  {n * 8 / f * 1e-12:.2f} TB/s (write)")
print(f"sum   {n * 8 / r * 1e-12:.2f} TB/s (read)")
print(f"amax  {n * 8 / m * 1e-12:.2f} TB/s (read)")
