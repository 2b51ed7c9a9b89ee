import torch, time, torch.nn.functional as F
dev="cuda:0"
M, D = 4096*51, 128
for dt in (torch.float16, torch.float32):
    x = torch.randn(M, D, device=dev, dtype=dt); w = torch.rand(D, device=dev, dtype=dt); b = torch.rand(D, device=dev, dtype=dt)
    def t(fn, n=20):
        for _ in range(3): fn()
        torch.cuda.synchronize(); t0=time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e6
    ref = F.layer_norm(x.float(), (D,), w.float(), b.float())
    def manual():
        var, mean = torch.var_mean(x, dim=-1, unbiased=False, keepdim=True)
        return torch.addcmul(b, (x - mean) * torch.rsqrt(var + 1e-5), w)
    def manual2():
        xf = x.view(-1, D)
        mean = xf.mean(-1, keepdim=True)
        xc = xf - mean
        rstd = torch.rsqrt((xc * xc).mean(-1, keepdim=True) + 1e-5)
        return torch.addcmul(b, xc * rstd, w)
    cands = {"layer_norm": lambda: F.layer_norm(x, (D,), w, b),
             "layer_norm 3d [4096,51,128]": lambda: F.layer_norm(x.view(4096, 51, D), (D,), w, b),
             "group_norm(1 group)": lambda: F.group_norm(x, 1, w, b),
             "group_norm [M/64,64,128]->(C=64?)": None,
             "instance_norm [1,M,128]": lambda: F.instance_norm(x.view(1, M, D)),
             "var_mean+addcmul": manual, "manual2": manual2,
             "native_layer_norm": lambda: torch.native_layer_norm(x, (D,), w, b, 1e-5)[0],
             "copy (x*1)": lambda: x * 1.0}
    for k, fn in cands.items():
        if fn is None: continue
        try:
            us = t(fn); err = (fn().float().view(M, D) - ref).abs().max().item() if k != "copy (x*1)" and "instance" not in k else float("nan")
            print(f"{dt} {k}: {us:.1f} us  err {err:.2e}", flush=True)
        except Exception as ex:
            print(f"{dt} {k}: FAILED {str(ex)[:150]}")
