import sys, time, torch
sys.path.insert(0, '/root/repo')
from mvuld_amd.graph import BatchedGraph
from mvuld_amd.data import synthetic
from mvuld_amd.graph import batch
for B in (32, 256):
    g = batch([synthetic.make_graph(i, 150, 250) for i in range(B)])
    t = time.perf_counter(); 
    for _ in range(5): g._index = None; g.index()
    th = (time.perf_counter() - t) / 5
    gd = BatchedGraph(g.src.cuda(), g.dst.cuda(), g.batch_num_nodes())
    gd.index(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t = time.perf_counter(); e0.record()
    for _ in range(5): gd._index = None; gd.index()
    e1.record(); torch.cuda.synchronize()
    print(f"B={B} nodes={g.number_of_nodes()} edges={g.num_edges()}: host {th*1e3:.2f} ms, device {e0.elapsed_time(e1)/5*1e3:.0f} us GPU / {(time.perf_counter()-t)/5*1e3:.2f} ms host-side")
