"""Micro-benchmark of the fused attention kernels per geometry (diagnostic)."""
import sys, os, math, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvuld_amd import ops, hip

dev = torch.device("cuda:0")
B = int(os.environ.get("B", 32))
cases = [("swin s0", 0, 4, 32, 112, 28, 14), ("swin s0 noshift", 0, 4, 32, 112, 28, 0), ("swin s1", 0, 8, 32, 56, 28, 14),
         ("swin s2", 0, 16, 32, 28, 28, 0), ("swin s3", 0, 32, 32, 14, 14, 0), ("roberta", 1, 12, 64, 0, 0, 0),
         ("roberta packed", 2, 12, 64, 0, 0, 0), ("roberta packed drop", 2, 12, 64, 0, 0, 0)]
if os.environ.get("GEOM"):            # extra window sizes (wave balance: a window of N tokens is N/16 tiles on 16 waves): GEOM=24,32
    cases += [(f"swin w{w}", 0, 16, 32, w, w, 0) for w in map(int, os.environ["GEOM"].split(","))]
which = sys.argv[1:] or ["auto"]
only = os.environ.get("CASES")
for name, mode, H, hd, res, ws, shift in cases:
    if only and only not in name:
        continue
    C = H * hd
    if mode == 0:
        N, nW = ws * ws, (res // ws) ** 2
        tokens = B * res * res
        g = ops.AttnGeom(0, B, H, hd, N, nW, res, ws, shift)
        T2 = (2 * ws - 1) ** 2
        table = torch.rand(T2, H, device=dev) * 16
        ls = torch.full((H,), 2.3, device=dev)
        valid = None
    elif mode == 1:
        N, nW = 512, 1
        tokens = B * N
        g = ops.AttnGeom(1, B, H, hd, N, 1, 0, 0, 0, 1 / math.sqrt(hd))
        table = ls = None
        valid = torch.ones(B, N, dtype=torch.int32, device=dev)
    else:                       # the fused step's text encoder: packed full rows (cu_seqlens), attention dropout 0.1 in training
        N, nW = 512, 1
        tokens = B * N
        g = ops.AttnGeom(2, B, H, hd, N, 1, tokens, 0, 0, 1 / math.sqrt(hd), sumsq=B * N * N, drop_p=0.1 if "drop" in name else 0.0, drop_seed=1234)
        table = ls = None
        valid = torch.arange(0, B + 1, dtype=torch.int32, device=dev) * N
    qkv = torch.randn(tokens, 3 * C, device=dev).to(torch.bfloat16)
    dout = torch.randn(tokens, C, device=dev).to(torch.bfloat16)
    pairs = B * nW * H * N * N
    for impl in which:
        ops.ATTN_IMPL[0] = impl
        dtab = torch.zeros((T2, H), device=dev) if mode == 0 else None
        dls = torch.zeros(H, device=dev) if mode == 0 else None
        for _ in range(2):
            out, lse = ops.attn_fwd(g, qkv, table, ls, valid)
            dq = ops.attn_bwd(g, qkv, out, dout, lse, table, ls, valid, dtab, dls)
        torch.cuda.synchronize()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        it = int(os.environ.get('IT', 5))
        e[0].record()
        for _ in range(it):
            out, lse = ops.attn_fwd(g, qkv, table, ls, valid)
        e[1].record()
        for _ in range(it):
            dq = ops.attn_bwd(g, qkv, out, dout, lse, table, ls, valid, dtab, dls)
        e[2].record()
        torch.cuda.synchronize()
        tf, tb = e[0].elapsed_time(e[1]) / it, e[1].elapsed_time(e[2]) / it
        print(f"{name:16s} {impl:7s} fwd {tf*1e3:8.1f} us  bwd {tb*1e3:8.1f} us   pairs {pairs/1e6:8.1f} M   fwd {pairs/tf/1e6:7.1f} Gpair/s  "
              f"fwd {4*pairs*hd/tf/1e9:6.1f} TF/s  bwd {10*pairs*hd/tb/1e9:6.1f} TF/s")
