"""Micro-benchmark of the bf16 MFMA GEMM on the step's shapes (diagnostic)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvuld_amd import ops, hip

dev = torch.device("cuda:0")
shapes = [("s0 qkv", 401408, 384, 128), ("s0 fc1", 401408, 512, 128), ("s0 fc2", 401408, 128, 512),
          ("s1 qkv", 100352, 768, 256), ("s1 fc1", 100352, 1024, 256), ("s1 fc2", 100352, 256, 1024),
          ("s2 qkv", 25088, 1536, 512), ("s2 proj", 25088, 512, 512), ("s2 fc1", 25088, 2048, 512), ("s2 fc2", 25088, 512, 2048),
          ("s3 fc1", 6272, 4096, 1024), ("s3 fc2", 6272, 1024, 4096),
          ("rob qkv", 16384, 2304, 768), ("rob out", 16384, 768, 768), ("rob fc1", 16384, 3072, 768), ("rob fc2", 16384, 768, 3072),
          ("sq 4096", 4096, 4096, 4096)]


def timeit(fn, it=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it


for name, M, N, K in shapes:
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = torch.randn(N, K, device=dev).to(torch.bfloat16) * 0.05
    wt = w.t().contiguous()
    dy = torch.randn(M, N, device=dev).to(torch.bfloat16)
    bias = torch.randn(N, device=dev)
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    dx = torch.empty(M, K, device=dev, dtype=torch.bfloat16)
    dw = torch.zeros(N, K, device=dev)
    fl = 2.0 * M * N * K
    t_f = timeit(lambda: ops.gemm_nt(x, w, out=y, bias=bias))
    t_d = timeit(lambda: ops.gemm_nt(dy, wt, out=dx))
    sk = ops.wgrad_splitk(N, K, M)
    dyT, xT = ops.transpose(dy), ops.transpose(x)
    t_w = timeit(lambda: ops.gemm_nt(dyT, xT, out=dw, out_mode=hip.OUT_ATOMIC, splitk=sk))
    t_t = timeit(lambda: (ops.transpose(dy), ops.transpose(x)))
    wp = torch.nn.Parameter(torch.zeros(N, K, device=dev)); wp.grad = torch.zeros(N, K, device=dev)
    bp = torch.nn.Parameter(torch.zeros(N, device=dev)); bp.grad = torch.zeros(N, device=dev)
    t_tn = timeit(lambda: ops.linear_wgrad(dy, x, wp, bp))
    tt = timeit(lambda: torch.matmul(x, w.t()))
    print(f"{name:9s} M={M:6d} N={N:4d} K={K:4d}  fwd {fl/t_f/1e9:6.0f} TF ({t_f*1e3:7.0f}us)  dgrad {fl/t_d/1e9:6.0f} TF  wgrad {fl/t_w/1e9:6.0f} TF (sk={sk}, {t_w*1e3:6.0f}us)"
          f"  transposes {t_t*1e3:6.0f}us  TN-wgrad+bias {fl/t_tn/1e9:6.0f} TF ({t_tn*1e3:6.0f}us) | torch(hipBLASLt) fwd {fl/tt/1e9:6.0f} TF")
