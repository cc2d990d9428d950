import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
from mvuld_amd import hip
from mvuld_amd.models.GraphModel import cross_entropy
sys.argv=[sys.argv[0]]
a=bench.parse(); dev=torch.device("cuda:0"); hip.LIB.load()
config, model, opt, sched, batch = bench.build(a, dev, 0)
g, images, ids, labels, lens = batch
N = int(os.environ.get("STEPS", 30))
for i in range(N):
    loss,_ = cross_entropy(model(g, images, ids, seq_lens=lens), labels); loss.backward(); opt.clip_grad_norm_(5.0); opt.step(); opt.zero_grad()
    if i in (2, 10, N // 2, N - 1):
        torch.cuda.synchronize()
        print(i, "loss", round(float(loss), 4), "allocated GB", round(torch.cuda.memory_allocated()/2**30,2), "peak", round(torch.cuda.max_memory_allocated()/2**30,2), "reserved", round(torch.cuda.memory_reserved()/2**30,2))
