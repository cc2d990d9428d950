"""Where the two streams of the fused step spend their time (diagnostic): GPU timestamps of the ends of each branch's forward
and backward, taken with events on the stream the branch runs on."""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from mvuld_amd import hip, ops
from mvuld_amd.models.GraphModel import cross_entropy

sys.argv = [sys.argv[0]]
a = bench.parse()
dev = torch.device("cuda:0")
hip.LIB.load()
config, model, opt, sched, batch = bench.build(a, dev, 0)
g, images, ids, labels, lens = batch
marks = {}


def mark(name):
    e = torch.cuda.Event(enable_timing=True)
    e.record(torch.cuda.current_stream())
    marks[name] = e


orig_swin, orig_text, orig_graph = model.swin.forward_features, model.unixcoder.get_xcode_vec, model.head.forward_graph
model.swin.forward_features = lambda x: (lambda r: (mark("fwd swin done"), r)[1])(orig_swin(x))
model.unixcoder.get_xcode_vec = lambda x, sl=None: (lambda r: (mark("fwd text done"), r)[1])(orig_text(x, sl))
model.head.forward_graph = lambda gg: (lambda r: (mark("fwd graph done"), r)[1])(orig_graph(gg))
import mvuld_amd.models.swin_transformer_v2 as _sw
import mvuld_amd.models.unixcoder as _ux
_np_bwd = _sw._NormPoolFn.backward
_sw._NormPoolFn.backward = staticmethod(lambda ctx, *g_: (mark("bwd swin starts (head join backward done)"), _np_bwd(ctx, *g_))[1])
ops.on_backward_done("swin", lambda: mark("bwd swin done"), key="diag")
ops.on_backward_done("swin.layers.2", lambda: mark("bwd swin stage2 done"), key="diag")
ops.on_backward_done("unixcoder", lambda: mark("bwd text done (side end)"), key="diag")


def step():
    mark("start")
    logits = model(g, images, ids, seq_lens=lens)
    loss, _ = cross_entropy(logits, labels)
    mark("fwd head done")
    loss.backward()
    mark("bwd returned (main)")
    # where the other streams stand once the host has enqueued the whole backward (events on THOSE streams; un-traced)
    for nm, st in (("weight-gradient stream drained", getattr(model, "_wg", None)), ("text stream drained", getattr(model, "_side", None)),
                   ("graph stream drained", getattr(model, "_gs", None))):
        if st is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record(st)
            marks[nm] = e
    opt.clip_grad_norm_(config.TRAIN.CLIP_GRAD)
    opt.step()
    opt.zero_grad()
    mark("step done")


for _ in range(4):
    step()
torch.cuda.synchronize()
base = marks["start"]
for k, e in sorted(marks.items(), key=lambda kv: base.elapsed_time(kv[1])):
    print(f"{base.elapsed_time(e):8.2f} ms  {k}")
