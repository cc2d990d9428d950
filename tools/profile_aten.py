"""List the torch (aten) ops one fused train step issues besides the HIP library calls (diagnostic: stray copies)."""
import sys, os, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from torch.profiler import profile, ProfilerActivity

args = types.SimpleNamespace(cfg=bench.CFG if isinstance(bench.CFG, str) else None, opts=[], dtype="bf16", batch=int(os.environ.get("B", 8)))
sys.argv = [sys.argv[0]]
a = bench.parse()
a.batch = args.batch
device = torch.device("cuda:0")
from mvuld_amd import hip
hip.LIB.load()
from mvuld_amd.models.GraphModel import cross_entropy
config, model, opt, sched, batch = bench.build(a, device, 0)
g, images, ids, labels, lens = batch


def step():
    logits = model(g, images, ids, seq_lens=lens)
    loss, _ = cross_entropy(logits, labels)
    loss.backward()
    opt.clip_grad_norm_(config.TRAIN.CLIP_GRAD)
    opt.step()
    opt.zero_grad()


step(); step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="count", row_limit=25, max_name_column_width=60))
print(prof.key_averages(group_by_stack_n=6).table(sort_by="count", row_limit=40, max_name_column_width=50, max_src_column_width=110))
