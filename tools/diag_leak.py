"""240 training steps of the bench configuration: loss, allocated / reserved / peak device memory at steps 20, 60, 120, 240 (diagnostic:
steady allocation = no per-step leak from the deferred-launch lists and the multi-stream record_stream bookkeeping)."""
import sys, os, json, torch
sys.path.insert(0, os.getcwd())
import bench
from mvuld_amd import hip, ops
from mvuld_amd.models.GraphModel import cross_entropy
sys.argv = [sys.argv[0]]
a = bench.parse()
dev = torch.device("cuda:0")
hip.LIB.load()
config, model, opt, sched, batch = bench.build(a, dev, 0)
g, images, ids, labels, lens = batch
def step(i):
    logits = model(g, images, ids, seq_lens=lens)
    loss, _ = cross_entropy(logits, labels)
    loss.backward()
    opt.clip_grad_norm_(config.TRAIN.CLIP_GRAD)
    opt.step(); opt.zero_grad(); sched.step_update(i)
    return loss
for i in range(241):
    l = step(i)
    if i in (20, 60, 120, 240):
        torch.cuda.synchronize()
        print(i, "loss", round(float(l), 5), "alloc MB", torch.cuda.memory_allocated() >> 20, "reserved MB", torch.cuda.memory_reserved() >> 20, "max MB", torch.cuda.max_memory_allocated() >> 20, flush=True)
