"""In-kernel stamps of the fused window backward (build_variants/libmvuld_afx9.so, -DAF_X=9): where one step of workgroup 0 spends its cycles."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MVULD_HIP_LIB"] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "build_variants", "libmvuld_afx9.so")
os.environ["MVULD_ATTN_BWD_FUSED"] = "1"
import torch
from mvuld_amd import ops, hip
dev = torch.device("cuda:0")
B, H, hd, res, ws, shift = 32, 16, 32, 28, 28, 0
C = H * hd
g = ops.AttnGeom(0, B, H, hd, ws * ws, 1, res, ws, shift)
T2 = (2 * ws - 1) ** 2
qkv = torch.randn(B * res * res, 3 * C, device=dev).to(torch.bfloat16)
dout = torch.randn(B * res * res, C, device=dev).to(torch.bfloat16)
table = torch.rand(T2, H, device=dev) * 16
ls = torch.full((H,), 2.3, device=dev)
out, lse = ops.attn_fwd(g, qkv, table, ls)
for _ in range(3):
    dtab, dls = torch.zeros((T2, H), device=dev), torch.zeros(H, device=dev)
    ops.attn_bwd(g, qkv, out, dout, lse, table, ls, None, dtab, dls)
torch.cuda.synchronize()
buf = (ctypes.c_longlong * 128)()
dll = hip.LIB.load()
dll.mvuld_debug_af_stamps.argtypes = [ctypes.c_void_p]
assert dll.mvuld_debug_af_stamps(buf) == 0
names = ["top", "q_issue", "frags", "blk0", "blk1", "blk2", "blk3", "blk4", "blk5", "blk6", "partial", "commit", "B1", "epilogue", "B2"]
for w in range(4):
    st = [buf[w * 32 + i] for i in range(15)]
    print(f"wave {w}: total {st[14] - st[0]} cycles | " + " ".join(f"{names[i]}+{st[i] - st[i - 1]}" for i in range(1, 15)))
    k = [buf[w * 32 + i] for i in range(16, 21)]
    print(f"        kernel phases: staging K/V/table {k[1] - k[0]}, first row fetch+commit {k[2] - k[1]}, {ws} steps {k[3] - k[2]} ({(k[3] - k[2]) // ws} per step), tail (chain flush, dK/dV epilogue) {k[4] - k[3]}, whole {k[4] - k[0]}")
