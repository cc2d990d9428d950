"""Stage-by-stage bf16-vs-fp32 comparison of the head on the GPU (diagnostic, not a test)."""
import sys, os, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from util import load_synth_into, synth, rel_l2
from mvuld_amd.models import GraphModel as GM
from mvuld_amd.data import synthetic
from mvuld_amd.graph import batch

dev = torch.device("cuda:0")
nodes = [60, 100, 130, 217]
g = batch([synthetic.make_graph(2000 + i, n, n) for i, n in enumerate(nodes)])
img = synth.tensor("head/img", (4, 1024), -1, 1)
txt = synth.tensor("head/txt", (4, 768), -1, 1)
cfg = types.SimpleNamespace(MODEL=types.SimpleNamespace(NUM_CLASSES=2))
mode = sys.argv[1] if len(sys.argv) > 1 else "train"
caps = {}
for dtype in (torch.float32, torch.bfloat16):
    m = GM.Multi_DefectModel_new_GCN(cfg, act_dtype=dtype)
    m.p_gat = m.p_mlp = m.p_hidden = 0.0
    m.gat.feat_drop_p = m.gat2.feat_drop_p = 0.0
    load_synth_into(m)
    m = m.to(dev).train(mode == "train")
    cap = {}
    orig_rs = GM.Rs_GCN.forward_rows
    def fr(self, v, B, _o=orig_rs, cap=cap):
        out, R = _o(self, v, B)
        cap[f"rs{len([k for k in cap if k.startswith('rs')])}"] = out.detach().float().cpu()
        return out, R
    GM.Rs_GCN.forward_rows = fr
    orig_l2 = GM._L2NormMeanFn.apply
    gg = g.to(dev)
    ig = img.to(dev).to(dtype).requires_grad_(True)
    tg = txt.to(dev).to(dtype).requires_grad_(True)
    lg = m(gg, ig, tg)
    cap["hgat"] = gg.ndata["HGATOUTPUT"].detach().float().cpu()
    cap["logits"] = lg.detach().float().cpu()
    loss, _ = GM.cross_entropy(lg, torch.tensor([0, 1, 1, 0], device=dev))
    loss.backward()
    cap["d_img"] = ig.grad.float().cpu(); cap["d_txt"] = tg.grad.float().cpu()
    for n, p in m.named_parameters():
        if p.grad is not None:
            cap["g:" + n] = p.grad.float().cpu()
    GM.Rs_GCN.forward_rows = orig_rs
    caps[dtype] = cap
a, b = caps[torch.float32], caps[torch.bfloat16]
for k in a:
    if k in b:
        print(f"{k:40s} rel_l2 = {rel_l2(b[k], a[k]):.3e}   |ref| = {float(a[k].norm()):.3e}")
