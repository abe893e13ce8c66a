import os, time, torch
n = 4096
torch.manual_seed(0)
A = torch.rand((n, n), dtype=torch.float64, device="cuda") - 0.5
B = torch.rand((n, n), dtype=torch.float64, device="cuda") - 0.5
for _ in range(3):
    C = A @ B
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    C = A @ B
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 10
ref = (A[:64].cpu().numpy().astype("longdouble") @ B.cpu().numpy().astype("longdouble"))
err = float(abs(C[:64].cpu().numpy() - ref).max() / abs(ref).max())
print(f"LD_PRELOAD={'yes' if 'gemmul8' in os.environ.get('LD_PRELOAD','') else 'no'}  torch DGEMM {n}^3: {2*n**3/dt*1e-12:.1f} TFLOPS, normwise err {err:.2e}")

# float32: PyTorch on ROCm sends these to hipBLASLt (hipblasLtMatmul) or hipBLAS (hipblasSgemm / GemmEx) depending on shape and settings
Af = (torch.rand((n, n), dtype=torch.float32, device="cuda") - 0.5)
Bf = (torch.rand((n, n), dtype=torch.float32, device="cuda") - 0.5)
Cf = Af @ Bf
torch.cuda.synchronize()
reff = Af[:64].double() @ Bf.double()
errf = float(((Cf[:64].double() - reff).abs().max() / reff.abs().max()).item())
print(f"torch SGEMM {n}^3: sgemm normwise err {errf:.2e}")

# float32 batched: torch.bmm -> hipblasLtMatmul with batched matrix layouts (or hipblasSgemmStridedBatched)
Xf = torch.rand((5, 640, 512), dtype=torch.float32, device="cuda") - 0.5
Yf = torch.rand((5, 512, 384), dtype=torch.float32, device="cuda") - 0.5
Zf = torch.bmm(Xf, Yf)
torch.cuda.synchronize()
errbf = 0.0
for i in range(5):
    refbf = Xf[i].double() @ Yf[i].double()
    errbf = max(errbf, float(((Zf[i].double() - refbf).abs().max() / refbf.abs().max()).item()))
print(f"torch.bmm float32 5 x 640x384x512: batched-f32 normwise err {errbf:.2e}")

# float32 torch.nn.Linear with a bias: hipblasLtMatmul with the BIAS epilogue
lin = torch.nn.Linear(512, 384, bias=True, device="cuda", dtype=torch.float32)
xl = torch.rand((640, 512), dtype=torch.float32, device="cuda") - 0.5
with torch.no_grad():
    yl = lin(xl)
    refl = xl.double() @ lin.weight.double().T + lin.bias.double()
torch.cuda.synchronize()
print(f"torch.nn.Linear float32 640x512 -> 384: linear-f32 normwise err {float(((yl.double() - refl).abs().max() / refl.abs().max()).item()):.2e}")

# batched: torch.bmm -> hipblasDgemmStridedBatched / hipblasGemmStridedBatchedEx
nb, b = 1024, 6
X = torch.rand((b, nb, nb), dtype=torch.float64, device="cuda") - 0.5
Y = torch.rand((b, nb, nb), dtype=torch.float64, device="cuda") - 0.5
Z = torch.bmm(X, Y)
torch.cuda.synchronize()
errb = 0.0
for i in range(b):   # every item: the hook spreads them over GEMMUL8_BATCH_STREAMS stream lanes
    refb = X[i, :64].cpu().numpy().astype("longdouble") @ Y[i].cpu().numpy().astype("longdouble")
    errb = max(errb, float(abs(Z[i, :64].cpu().numpy() - refb).max() / abs(refb).max()))
print(f"torch.bmm {b} x {nb}^3: bmm normwise err {errb:.2e}")
for nbb, bb in ((512, 32), (1024, 16), (2048, 8)):
    X2 = torch.rand((bb, nbb, nbb), dtype=torch.float64, device="cuda") - 0.5
    for _ in range(2):
        Z2 = torch.bmm(X2, X2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        Z2 = torch.bmm(X2, X2)
    torch.cuda.synchronize()
    dtb = (time.perf_counter() - t0) / 5
    print(f"torch.bmm {bb} x {nbb}^3 (lanes={os.environ.get('GEMMUL8_BATCH_STREAMS', 'default')}): {2 * bb * nbb**3 / dtb * 1e-12:.1f} TFLOPS")

# HIP-graph capture of a hooked matmul (torch.cuda.graph warms nothing up by itself: run once on a side stream first)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    Cg = A @ B
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph, stream=s):
    Cg = A @ B
A.mul_(2.0)
graph.replay()
torch.cuda.synchronize()
Ce = A @ B
torch.cuda.synchronize()
print(f"graph replay equals eager: {bool(torch.equal(Cg, Ce))}")
