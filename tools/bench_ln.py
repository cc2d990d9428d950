"""Micro-benchmark of the LayerNorm kernels on the step's shapes (diagnostic): GB/s of algorithmic traffic."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvuld_amd import ops

dev = torch.device("cuda:0")


def timeit(fn, it=10):
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


SETS = int(os.environ.get("LN_SETS", "1"))      # > 1: rotate over that many operand sets, so a 256 MB last-level cache cannot hold them (in-step behaviour)
for rows, C in [(401408, 128), (100352, 256), (25088, 512), (6272, 1024), (16384, 768)]:
    xs = [torch.randn(rows, C, device=dev).to(torch.bfloat16) for _ in range(SETS)]
    rs_ = [torch.randn(rows, C, device=dev).to(torch.bfloat16) for _ in range(SETS)]
    x, r = xs[0], rs_[0]
    g = torch.nn.Parameter(torch.rand(C, device=dev)); g.grad = torch.zeros(C, device=dev)
    b = torch.nn.Parameter(torch.rand(C, device=dev)); b.grad = torch.zeros(C, device=dev)
    y, mean, rstd, _ = ops.layernorm_fwd(x, g, b)
    nb = rows * C * 2
    k = [0]

    def rot():
        k[0] = (k[0] + 1) % SETS
        return xs[k[0]], rs_[k[0]]
    t0 = timeit(lambda: ops.layernorm_fwd(rot()[0], g, b))
    t1 = timeit(lambda: ops.layernorm_fwd(*rot()[:1], g, b, residual=rs_[k[0]]))
    t2 = timeit(lambda: ops.layernorm_fwd(rot()[0], g, b, pre=rs_[k[0]], want_sum=True))
    t3 = timeit(lambda: ops.layernorm_bwd(rot()[1], xs[k[0]], g, b, mean, rstd))
    tc = timeit(lambda: x.clone())
    print(f"rows={rows:6d} C={C:4d}  fwd {2*nb/t0/1e6:6.0f} GB/s ({t0*1e3:5.0f}us)  fwd+res {3*nb/t1/1e6:6.0f} GB/s ({t1*1e3:5.0f}us)"
          f"  fwd+pre+sum {4*nb/t2/1e6:6.0f} GB/s ({t2*1e3:5.0f}us)  bwd {3*nb/t3/1e6:6.0f} GB/s ({t3*1e3:5.0f}us)  torch clone {2*nb/tc/1e6:6.0f} GB/s")
