#!/usr/bin/env python3
"""Per-op timing of one attention block of the stand-in policy at rollout batch size (developer tool, GPU box)."""
import sys, time, torch, torch.nn.functional as F
dev = "cuda:0"; dt = torch.float16
B, N, D, h = 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 51, 128, 8
dk = D // h
x = torch.randn(B, N, D, device=dev, dtype=dt)
wqkv = torch.randn(D, 3 * D, device=dev, dtype=dt) * 0.05
wq, wk, wv = (wqkv[:, i * D:(i + 1) * D].contiguous() for i in range(3))
wo = torch.randn(D, D, device=dev, dtype=dt) * 0.05
vw = torch.randn(1024, D, device=dev, dtype=dt) * 0.05
w2 = torch.randn(D, 512, device=dev, dtype=dt) * 0.05
g = torch.ones(D, device=dev, dtype=dt); b = torch.zeros(D, device=dev, dtype=dt)

def t(name, fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); print(f"{name:58s} {(time.perf_counter() - t0) / n * 1e6:8.1f} us", flush=True)

x2 = x.view(-1, D)
t("qkv fused GEMM [M,128]x[128,384]", lambda: x2 @ wqkv)
t("3 separate GEMMs [M,128]x[128,128]", lambda: (x2 @ wq, x2 @ wk, x2 @ wv))
qkv = x2 @ wqkv
Qv, Kv, Vv = qkv.view(B, N, 3, h, dk).permute(2, 0, 3, 1, 4)
t("SDPA on views of the fused projection", lambda: F.scaled_dot_product_attention(Qv, Kv, Vv))
Qc, Kc, Vc = (z.contiguous() for z in (Qv, Kv, Vv))
t("SDPA on contiguous [B,h,N,dk]", lambda: F.scaled_dot_product_attention(Qc, Kc, Vc))
Qt, Kt, Vt = ((x2 @ w).view(B, N, h, dk).transpose(1, 2) for w in (wq, wk, wv))
t("SDPA on [B,N,h,dk]-layout transposed views", lambda: F.scaled_dot_product_attention(Qt, Kt, Vt))
o = F.scaled_dot_product_attention(Qt, Kt, Vt)
print("  out strides", o.stride(), "contig after transpose(1,2):", o.transpose(1, 2).is_contiguous())
t("out.transpose(1,2).reshape", lambda: o.transpose(1, 2).reshape(B * N, D))
heads = o.transpose(1, 2).reshape(B * N, D)
t("addmm(x, heads, wo)", lambda: torch.addmm(x2, heads, wo))
t("group_norm", lambda: F.group_norm(x2, 1, g, b, 1e-5))
t("FFN up GEMM [M,128]x[128,1024]", lambda: x2 @ vw.t())
hid = x2 @ vw.t()
t("glu", lambda: F.glu(hid, dim=-1))
gl = F.glu(hid, dim=-1)
t("addmm(x, glu, w2^T)", lambda: torch.addmm(x2, gl, w2.t()))
# math attention for comparison
def math_attn():
    s = torch.matmul(Qc, Kc.transpose(2, 3)) * (dk ** -0.5)
    return torch.matmul(torch.softmax(s, -1), Vc)
t("math attention (matmul, softmax, matmul) contiguous", math_attn)
# single-query attention (global decoders): Nq = 1
q1 = torch.randn(B, h, 1, dk, device=dev, dtype=dt)
t("SDPA Nq=1 vs Nk=N", lambda: F.scaled_dot_product_attention(q1, Kc, Vc))
def one_query():
    s = (q1 * Kc).sum(-1) * (dk ** -0.5)                 # B,h,N
    return (torch.softmax(s, -1).unsqueeze(-1) * Vc).sum(2)
t("elementwise single-query attention", one_query)
t("kv projection for Nq=1 block [M,128]x[128,256]", lambda: x2 @ wqkv[:, D:])
# re-associated single-query attention: no K / V projection of the Nk tokens
Wk3 = wqkv[:, D:2 * D].reshape(D, h, dk); Wv3 = wqkv[:, 2 * D:].reshape(D, h, dk)
qh = q1[:, :, 0, :]                                   # B,h,dk
def reassoc():
    U = torch.einsum('bhk,dhk->bhd', qh, Wk3)         # B,h,D
    s = torch.bmm(U, x.transpose(1, 2)) * (dk ** -0.5)   # B,h,N
    a = torch.softmax(s, -1)
    ctx = torch.bmm(a, x)                              # B,h,D
    return torch.einsum('bhd,dhk->bhk', ctx, Wv3).reshape(B, D)
t("re-associated single-query attention (incl. no kv proj)", reassoc)
t("  bmm(U, x^T)", lambda: torch.bmm(torch.einsum('bhk,dhk->bhd', qh, Wk3), x.transpose(1, 2)))
a_ = torch.softmax(torch.bmm(torch.einsum('bhk,dhk->bhd', qh, Wk3), x.transpose(1, 2)), -1)
t("  bmm(a, x)", lambda: torch.bmm(a_, x))
ref = F.scaled_dot_product_attention(q1, *((x2 @ wqkv[:, D:]).view(B, N, 2, h, dk).permute(2, 0, 3, 1, 4))).reshape(B, D)
print("  max abs diff vs SDPA path:", (reassoc() - ref).abs().max().item(), "ref scale", ref.abs().max().item())
