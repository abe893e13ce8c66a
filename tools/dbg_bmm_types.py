import torch, time
for dt in (torch.float32, torch.float64, torch.complex64):
    X = torch.rand((8, 512, 512), dtype=torch.float32, device="cuda").to(dt)
    Y = torch.rand((8, 512, 512), dtype=torch.float32, device="cuda").to(dt)
    Z = torch.bmm(X, Y); torch.cuda.synchronize()
    ref = (X[3].to(torch.complex128 if dt.is_complex else torch.float64) @ Y[3].to(torch.complex128 if dt.is_complex else torch.float64))
    err = float(((Z[3].to(ref.dtype) - ref).abs().max() / ref.abs().max()).item())
    print(dt, "bmm err", err)
    W = torch.matmul(X, Y[0]); torch.cuda.synchronize()     # broadcast: batched x single
    Xt = X.transpose(1, 2); V = torch.bmm(Xt, Y); torch.cuda.synchronize()
    print(dt, "done")
