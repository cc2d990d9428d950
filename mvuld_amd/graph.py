"""Batched code-property-graph container: what the head needs from ``dgl.DGLGraph``.

The reference hands the head a ``dgl.batch``-ed graph (bigvul_dataset.py:177-205)
and touches only ``g.ndata[...]``, ``g.to(device)`` and ``dgl.unbatch``
(GraphModel.py:30-54,163-181).  This container keeps the same surface and adds
the index structures the HIP kernels consume:

  * CSR by destination (``indptr_dst``/``src_by_dst``) -- the incoming-edge list the
    GAT edge-softmax + aggregation kernel walks (one wave per (dst, head));
  * CSR by source (``indptr_src``/``dst_by_src``/``eid_by_src``) -- the transposed
    index that turns the backward scatter into a gather;
  * ``node_offsets`` -- per-graph node ranges for the pad/truncate-to-100 kernel.

Multi-edges and repeated self-loops are kept: each is its own softmax term
(data_list.py:279,314 build ``dgl.graph((eo, ei))`` then ``add_self_loop`` with
no dedup).  Indices are int32 on device (int64 in DGL).
"""
from typing import Dict, List, Optional

import torch


class BatchedGraph:
    def __init__(self, src: torch.Tensor, dst: torch.Tensor, batch_num_nodes, ndata: Optional[Dict] = None,
                 edata: Optional[Dict] = None):
        self.src = src.to(torch.int64)
        self.dst = dst.to(torch.int64)
        bnn = torch.as_tensor(batch_num_nodes, dtype=torch.int64)
        self._batch_num_nodes = bnn.cpu()
        self.ndata: Dict[str, torch.Tensor] = dict(ndata or {})
        self.edata: Dict[str, torch.Tensor] = dict(edata or {})
        self._n = int(self._batch_num_nodes.sum())
        if self.src.numel():
            assert int(self.src.max()) < self._n and int(self.dst.max()) < self._n, "edge endpoint out of range"
            assert int(self.src.min()) >= 0 and int(self.dst.min()) >= 0
        self._index = None

    # ---- dgl-like surface -------------------------------------------------
    def number_of_nodes(self):
        return self._n

    num_nodes = number_of_nodes

    def num_edges(self):
        return int(self.src.numel())

    def batch_num_nodes(self):
        return self._batch_num_nodes

    @property
    def batch_size(self):
        return int(self._batch_num_nodes.numel())

    @property
    def device(self):
        return self.src.device

    def to(self, device, non_blocking=False):
        g = BatchedGraph.__new__(BatchedGraph)
        g.src = self.src.to(device, non_blocking=non_blocking)
        g.dst = self.dst.to(device, non_blocking=non_blocking)
        g._batch_num_nodes = self._batch_num_nodes
        g.ndata = {k: v.to(device, non_blocking=non_blocking) for k, v in self.ndata.items()}
        g.edata = {k: v.to(device, non_blocking=non_blocking) for k, v in self.edata.items()}
        g._n = self._n
        g._index = None
        if self._index is not None:
            g._index = {k: v.to(device, non_blocking=non_blocking) for k, v in self._index.items()}
        return g

    def _index_on_device(self):
        """The same structures built by the library on the graph's device (mvuld_graph_csr_build: stable radix sorts + gathers): no
        device -> host round trip when a graph reaches the GPU without an index."""
        from . import hip
        from .hip import call, ptr
        n, e, dev = self._n, self.num_edges(), self.src.device
        src = self.src.to(torch.int64).contiguous()
        dst = self.dst.to(torch.int64).contiguous()
        out = {k: torch.empty(n + 1 if k.startswith("indptr") else e, dtype=torch.int32, device=dev)
               for k in ("indptr_dst", "src_by_dst", "indptr_src", "dst_by_src", "slot_by_src")}
        nb = hip.LIB.fn("mvuld_graph_csr_workspace_bytes")(e)
        ws = torch.empty(nb, dtype=torch.uint8, device=dev)
        call("graph_csr_build", ptr(src), ptr(dst), e, n, ptr(out["indptr_dst"]), ptr(out["src_by_dst"]), ptr(out["indptr_src"]),
             ptr(out["dst_by_src"]), ptr(out["slot_by_src"]), ptr(ws), nb)
        off = torch.zeros(self.batch_size + 1, dtype=torch.int64)
        off[1:] = torch.cumsum(self._batch_num_nodes, 0)
        out["node_offsets"] = off.to(torch.int32).to(dev, non_blocking=True)
        return out

    # ---- index structures for the kernels -------------------------------
    def index(self):
        """dict of int32 tensors on the graph's device (built once, cached)."""
        if self._index is None and self.src.is_cuda and self.num_edges() > 0:
            self._index = self._index_on_device()
        if self._index is None:
            n = self._n
            src, dst = self.src.cpu(), self.dst.cpu()
            e = src.numel()
            eid = torch.arange(e, dtype=torch.int64)
            # by destination (stable => per-dst edge order == edge-id order)
            order_d = torch.sort(dst, stable=True).indices
            cnt_d = torch.bincount(dst, minlength=n)
            indptr_d = torch.zeros(n + 1, dtype=torch.int64)
            indptr_d[1:] = torch.cumsum(cnt_d, 0)
            # position of every edge inside the by-dst ordering
            pos_in_d = torch.empty(e, dtype=torch.int64)
            pos_in_d[order_d] = eid
            order_s = torch.sort(src, stable=True).indices
            cnt_s = torch.bincount(src, minlength=n)
            indptr_s = torch.zeros(n + 1, dtype=torch.int64)
            indptr_s[1:] = torch.cumsum(cnt_s, 0)
            off = torch.zeros(self.batch_size + 1, dtype=torch.int64)
            off[1:] = torch.cumsum(self._batch_num_nodes, 0)
            idx = {
                "indptr_dst": indptr_d, "src_by_dst": src[order_d],
                "indptr_src": indptr_s, "dst_by_src": dst[order_s],
                # for an edge listed in by-src order: its slot in the by-dst ordering
                "slot_by_src": pos_in_d[order_s],
                "node_offsets": off,
            }
            self._index = {k: v.to(torch.int32).to(self.src.device) for k, v in idx.items()}
        return self._index


def batch(graphs: List[BatchedGraph]) -> BatchedGraph:
    """Concatenate graphs, offsetting node ids (dgl.batch)."""
    srcs, dsts, bnn, off = [], [], [], 0
    for g in graphs:
        srcs.append(g.src + off)
        dsts.append(g.dst + off)
        bnn.extend(g.batch_num_nodes().tolist())
        off += g.number_of_nodes()
    keys = graphs[0].ndata.keys()
    nd = {k: torch.cat([g.ndata[k] for g in graphs], 0) for k in keys}
    ed = {k: torch.cat([g.edata[k] for g in graphs], 0) for k in graphs[0].edata.keys()}
    return BatchedGraph(torch.cat(srcs), torch.cat(dsts), bnn, nd, ed)


def unbatch(g: BatchedGraph) -> List[BatchedGraph]:
    """Split back into single graphs (dgl.unbatch); edges must not cross graphs."""
    out, off = [], 0
    src, dst = g.src, g.dst
    for n in g.batch_num_nodes().tolist():
        m = (dst >= off) & (dst < off + n)
        nd = {k: v[off:off + n] for k, v in g.ndata.items()}
        ed = {k: v[m] for k, v in g.edata.items()}
        out.append(BatchedGraph(src[m] - off, dst[m] - off, [n], nd, ed))
        off += n
    return out


def add_self_loop(g: BatchedGraph) -> BatchedGraph:
    """Append one self-loop per node without dedup (dgl.add_self_loop; data_list.py:314)."""
    n = g.number_of_nodes()
    loops = torch.arange(n, dtype=torch.int64, device=g.src.device)
    ed = {}
    for k, v in g.edata.items():
        ed[k] = torch.cat([v, torch.zeros(n, *v.shape[1:], dtype=v.dtype, device=v.device)], 0)
    return BatchedGraph(torch.cat([g.src, loops]), torch.cat([g.dst, loops]), g.batch_num_nodes(), g.ndata, ed)
