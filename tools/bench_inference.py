"""BASELINE config 4 (diagnostic, not the headline): inference-only fused forward captured in a hipGraph (torch.cuda.CUDAGraph
over the library's launches on both streams), replayed; prints eager vs graph time and the max logit difference.  B=<batch>."""
import os, sys, types, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(sys.path[0], "tests"))
import torch
from util import load_synth_into
from mvuld_amd.config import get_config
from mvuld_amd.main_bigvul import build_fused_model
from mvuld_amd.data import synthetic
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = os.path.join(root, "mvuld_amd", "configs", "mySwin", "swinv2_base_patch4_window24to28_384to448_1ktoMYDATA_ft.yaml")
config = get_config(types.SimpleNamespace(cfg=cfg, opts=["FUSED.DTYPE", "bf16"], batch_size=8, local_rank=0))
gpu = torch.device("cuda:0")
model = build_fused_model(config); load_synth_into(model); model = model.to(gpu).eval()
f = config.FUSED
B = int(os.environ.get("B", 8))
g, images, ids, _ = synthetic.make_batch(list(range(70, 70 + B)), config.DATA.IMG_SIZE, f.SEQ_LEN, f.TEXT.VOCAB, f.NODES_LO, f.NODES_HI)
g = g.to(gpu); g.index(); images = images.to(gpu); ids = ids.to(gpu)
with torch.no_grad():
    for _ in range(2):
        eager = model(g, images, ids).float().clone()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): model(g, images, ids)
    torch.cuda.synchronize(); te = (time.perf_counter() - t0) / 5
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        model(g, images, ids)
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = model(g, images, ids)
    graph.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): graph.replay()
    torch.cuda.synchronize(); tg = (time.perf_counter() - t0) / 5
    print("max diff", float((out.float() - eager).abs().max()), "eager ms", te * 1e3, "graph ms", tg * 1e3, "functions/s graph", B / tg)
