"""Fused MLP (csrc/mlp_panel.hip) against the unfused GEMM launches on the Swin stage-0 / stage-1 shapes at batch 32 (diagnostic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvuld_amd import ops, hip

dev = torch.device("cuda:0")
bf = torch.bfloat16
for name, M, C in (("s0", 32 * 12544, 128), ("s1", 32 * 3136, 256)):
    x = torch.randn(M, C, device=dev).to(bf)
    dy = torch.randn(M, C, device=dev).to(bf)
    g = torch.randn(M, C, device=dev).to(bf)
    W1 = torch.nn.Parameter(torch.randn(4 * C, C, device=dev) * C ** -0.5)
    B1 = torch.nn.Parameter(torch.randn(4 * C, device=dev) * 0.1)
    W2 = torch.nn.Parameter(torch.randn(C, 4 * C, device=dev) * (4 * C) ** -0.5)
    B2 = torch.nn.Parameter(torch.randn(C, device=dev) * 0.1)
    hpre = torch.empty((M, 4 * C), dtype=bf, device=dev)

    def unf_f():
        h = ops.gemm_nt(x, ops.weight(W1, bf), bias=B1.data, epi=hip.EPI_GELU, aux=hpre)
        return h, ops.gemm_nt(h, ops.weight(W2, bf), bias=B2.data)

    def unf_b():
        dh = ops.gemm_nt(dy, ops.weight_t(W2, bf), epi=hip.EPI_MUL_DGELU, aux=hpre)
        return dh, ops.gemm_nt(dh, ops.weight_t(W1, bf), epi=hip.EPI_ADD_AUX, aux=g)
    fns = {"fused fwd": lambda: ops.mlp_fused_fwd(x, W1, B1, W2, B2), "unfused fwd": unf_f,
           "fused bwd": lambda: ops.mlp_fused_bwd(x, dy, g, W1, B1, W2), "unfused bwd": unf_b}
    for k, fn in fns.items():
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record()
        torch.cuda.synchronize()
        print(f"{name} M={M} C={C} {k:12s} {e0.elapsed_time(e1) / 5 * 1e3:8.1f} us", flush=True)
