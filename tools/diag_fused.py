"""Diagnostic: fused window backward vs the three-pass kernels -- where do they differ?  (dev tool)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvuld_amd import ops, hip

dev = torch.device("cuda:0")
torch.manual_seed(0)
for (res, ws, shift, H) in [(28, 28, 0, 1), (56, 28, 14, 1), (16, 8, 4, 1), (8, 4, 0, 1)]:
    B, hd = 1, 32
    C = H * hd
    T2 = (2 * ws - 1) ** 2
    W2 = 2 * ws - 1
    qkv = torch.randn(B * res * res, 3 * C, device=dev).to(torch.bfloat16)
    dout = torch.randn(B * res * res, C, device=dev).to(torch.bfloat16)
    table = torch.rand(T2, H, device=dev) * 16
    ls = torch.full((H,), 2.3, device=dev)
    g = ops.AttnGeom(0, B, H, hd, ws * ws, (res // ws) ** 2, res, ws, shift)
    out, lse = ops.attn_fwd(g, qkv, table, ls)
    got = {}
    for fused in (1, 0):
        hip.LIB.fn("mvuld_set_attn_bwd_fused")(fused)
        dtab, dls = torch.zeros((T2, H), device=dev), torch.zeros(H, device=dev)
        dq = ops.attn_bwd(g, qkv, out, dout, lse, table, ls, None, dtab, dls)
        torch.cuda.synchronize()
        got[fused] = (dq.float(), dtab, dls)
    hip.LIB.fn("mvuld_set_attn_bwd_fused")(1)
    a, b = got[1], got[0]
    print(f"== res {res} ws {ws} shift {shift}")
    for nm, lo in (("dq", 0), ("dk", C), ("dv", 2 * C)):
        e = (a[0][:, lo:lo + C] - b[0][:, lo:lo + C]).abs().max() / b[0][:, lo:lo + C].abs().max()
        print(f"  {nm} rel {float(e):.4f}")
    ta, tb = a[1][:, 0].view(W2, W2), b[1][:, 0].view(W2, W2)
    err = (ta - tb).abs() / tb.abs().max()
    print(f"  dtab rel {float(err.max()):.4f}  dls {float(a[2][0]):.4f} vs {float(b[2][0]):.4f}")
    rows = err.max(1).values
    cols = err.max(0).values
    print("  per dy row err:", " ".join(f"{float(v):.2f}" for v in rows))
    print("  per dx col err:", " ".join(f"{float(v):.2f}" for v in cols))
    wr = int(rows.argmax())
    print(f"  worst row dy={wr - ws + 1}: fused", " ".join(f"{float(v):.3g}" for v in ta[wr]), "\n      3pass", " ".join(f"{float(v):.3g}" for v in tb[wr]))
