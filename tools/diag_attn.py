import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvuld_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
for (res, ws, shift, H, lsv) in [(56, 28, 14, 2, [3.9, 1.7]), (56, 28, 0, 2, [3.9, 1.7]), (56, 28, 14, 2, [2.3, 2.3])]:
    B, hd = 2, 32
    C = H * hd
    N, nW = ws * ws, (res // ws) ** 2
    qkv = (torch.rand(B * res * res, 3 * C, device=dev) * 4 - 2).to(torch.bfloat16)
    dout = (torch.rand(B * res * res, C, device=dev) * 2 - 1).to(torch.bfloat16)
    T2 = (2 * ws - 1) ** 2
    table = torch.rand(T2, H, device=dev) * 16
    ls = torch.tensor(lsv, device=dev)
    g = ops.AttnGeom(0, B, H, hd, N, nW, res, ws, shift)
    res_ = {}
    for impl in ("simple", "auto"):
        ops.ATTN_IMPL[0] = impl
        out, lse = ops.attn_fwd(g, qkv, table, ls)
        dtab = torch.zeros(T2, H, device=dev); dls = torch.zeros(H, device=dev)
        dq = ops.attn_bwd(g, qkv, out, dout, lse, table, ls, None, dtab, dls)
        res_[impl] = (out.float(), lse, dq.float(), dtab, dls)
    a, b = res_["simple"], res_["auto"]
    names = ["out", "lse", "dqkv", "dtab", "dls"]
    print((res, ws, shift, H, lsv), " ".join(f"{n}: {float((x-y).abs().max()/(x.abs().max()+1e-9)):.3e}" for n, x, y in zip(names, a, b)), "dls", a[4].tolist(), b[4].tolist())
    Cc = C
    d3 = (a[2] - b[2]).abs()
    print("   dq/dk/dv max err:", float(d3[:, :Cc].max()), float(d3[:, Cc:2*Cc].max()), float(d3[:, 2*Cc:].max()), " ref max", float(a[2][:, :Cc].abs().max()), float(a[2][:, Cc:2*Cc].abs().max()), float(a[2][:, 2*Cc:].abs().max()))
