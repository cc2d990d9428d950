"""The head's graph branch alone (forward_graph + its backward) on the bench batch: wall time per call and, under rocprofv3 --kernel-trace --stats,
its kernel list (diagnostic: the branch runs on its own stream in the step, but its ~400 small launches cost the step ~3 ms)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from mvuld_amd import hip, ops

sys.argv = [sys.argv[0]]
a = bench.parse()
dev = torch.device("cuda:0")
hip.LIB.load()
config, model, opt, sched, batch = bench.build(a, dev, 0)
g = batch[0]
head = model.head
model.train()
w = None
def once():
    global w
    h = head.forward_graph(g)
    if w is None:
        w = torch.randn_like(h)
    (h * w).sum().backward()
for _ in range(3):
    once()
torch.cuda.synchronize()
n = int(os.environ.get("IT", 10))
t0 = time.perf_counter()
for _ in range(n):
    once()
host = (time.perf_counter() - t0) / n * 1e3
torch.cuda.synchronize()
print(f"graph branch fwd+bwd: {(time.perf_counter() - t0) / n * 1e3:.2f} ms per call (host enqueue {host:.2f} ms)")
