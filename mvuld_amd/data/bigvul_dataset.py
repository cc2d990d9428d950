"""``bigvul_loader_graph(config)`` with the reference's return signature (mvuld/data/bigvul_dataset.py:157-216):
``train_data, val_data, test_data, loader_train, loader_val, loader_test, mixup_fn``.

Two sources.  ``FUSED.DATA_ROOT`` set: ``BigVulFiles`` reads a corpus directory in the reference's file formats (image list, PNGs,
Joern exports, OCR position pickles, cached features: data_list.py:73-153, 265-317) and drives ``joern_ingest`` + the device-side image
transform + on-device node embeddings.  Unset (the default: the Big-Vul corpus, the Joern graphs and the rendered PNGs are an external
download, README.md:32, not on this box): samples come from ``data.synthetic`` with the schema ``ImageList.__getitem__`` yields
(data_list.py:107-141).  Sharding is the reference's either way: ``DistributedSampler`` per split, ``shuffle=True, drop_last=True`` for
train (bigvul_dataset.py:163-185), graphs collated with ``graph.batch`` (dgl.batch).

Fused mode batch:  (g, images [B,3,S,S] f32, source_ids [B,L] i64, target [B] i64)
Head-only batch:   (g, img_embedding [B,1024], func_text_embedding [B,768], target)   (reference-faithful step)
"""
import torch
from torch.utils.data import DataLoader, Dataset
from torch.utils.data.distributed import DistributedSampler

from . import synthetic
from .. import synth
from ..graph import batch as batch_graphs


class SyntheticBigVul(Dataset):
    def __init__(self, n, base_index, config, fused=True):
        self.n, self.base, self.fused = n, base_index, fused
        f = config.FUSED
        self.img_size, self.seq_len = config.DATA.IMG_SIZE, f.SEQ_LEN
        self.vocab, self.n_lo, self.n_hi = f.TEXT.VOCAB, f.NODES_LO, f.NODES_HI

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        idx = self.base + i
        g = synthetic.make_graph(idx, self.n_lo, self.n_hi)
        label = synthetic.make_label(idx)
        if self.fused:
            return g, synthetic.make_image(idx, self.img_size), synthetic.make_ids(idx, self.seq_len, self.vocab), label
        return g, synth.tensor(f"imgfeat/{idx}", (1024,)), synth.tensor(f"txtfeat/{idx}", (768,)), label


class BigVulFiles(Dataset):
    """One split of a Big-Vul corpus directory, in the file formats the reference reads (all paths relative to ``root``):

        <split>.txt                      one line per function: "<png path> <label>"   (data_list.py:38-47 make_dataset; the id is the
                                         PNG's stem, :123)
        <png path>                       the rendered graph image, any size (decoded here, resized / normalised on the device)
        func_before/<id>.c.nodes.json    Joern export of the function (joern.py:260-268 next to ImageList.itempath, :228-232)
        func_before/<id>.c.edges.json
        norm_pos_dict/<id>.pkl           {line number: [x0, x1, y0, y1]} OCR boxes, normalised (data_list.py:146-153)
        token_ids/<id>.npy               int64 [<= SEQ_LEN] UniXcoder ids of the function text (fused mode; the tokeniser itself is out of
                                         scope: ids are prepared offline, as the reference's cache_g_items_tokenids does, :234-262)
        line_token_ids/<id>.npz          ``lineno`` int [m], ``ids`` int64 [m, Lk]: ids of every source line (node features are then
                                         computed on the device by the text encoder: FusedMVulD.forward(node_ids=...))
        node_emb/<id>.npz                (instead) ``lineno`` [m], ``emb`` f32 [m, 768]: cached per-line embeddings
        swinv2_method_level_try5/<id>.pt, result.pkl   head-only mode: cached image feature [1024] (:133) and the pandas frame with
                                         columns ``ids`` / ``repr`` of function embeddings [768] (:101-103, :138-139)

    ``__getitem__`` -> fused: (g, image uint8 [H, W, 3], source_ids [SEQ_LEN], label); head-only: (g, img_embedding, func_embedding,
    label) -- the tuple ImageList.__getitem__ returns (:141).  The graph is built from the Joern export exactly as ImageList.item does
    (:265-317 via joern_ingest.build_function_graph): line-level nodes, AST / CFG / CDG edges, OCR boxes, one self-loop per node."""

    def __init__(self, root, split, config, fused=True, graph_type="all"):
        import os
        self.root, self.fused, self.graph_type = str(root), fused, graph_type
        self.seq_len = config.FUSED.SEQ_LEN
        # width every graph's per-line ids are padded (pad id 1) or truncated to, so that functions whose npz files hold different Lk
        # batch together (graph.batch concatenates ndata along dim 0)
        self.line_len = int(config.FUSED.LINE_LEN)
        self._node_source = None          # "line_token_ids" | "node_emb": one node-feature source per corpus (graph.batch keys by the first graph)
        self.items = []
        with open(os.path.join(self.root, f"{split}.txt")) as f:
            for line in f:
                parts = line.split()
                if len(parts) >= 2:
                    self.items.append((parts[0], int(parts[1])))
        self._func_emb = None

    def __len__(self):
        return len(self.items)

    def _path(self, *parts):
        import os
        p = os.path.join(*parts)
        return p if os.path.isabs(p) else os.path.join(self.root, p)

    def graph(self, _id):
        import os
        import pickle
        import numpy as np
        from .joern_ingest import build_function_graph, load_cpg
        nodes_json, edges_json = load_cpg(self._path("func_before", f"{_id}.c"))
        pos_file = self._path("norm_pos_dict", f"{_id}.pkl")
        pos = None
        if os.path.exists(pos_file):
            with open(pos_file, "rb") as f:
                pos = pickle.load(f)
        g, code = build_function_graph(nodes_json, edges_json, pos, self.graph_type)
        lineno = g.ndata["_lineno"].to(torch.int64).tolist()
        lt, ne = self._path("line_token_ids", f"{_id}.npz"), self._path("node_emb", f"{_id}.npz")
        source = "line_token_ids" if os.path.exists(lt) else ("node_emb" if os.path.exists(ne) else None)
        if source is not None and self._node_source not in (None, source):
            raise ValueError(f"{_id}: node features come from {source}/ but earlier functions of this corpus used {self._node_source}/ -- "
                             f"a corpus must use ONE node-feature source (graphs of a batch share their ndata keys)")
        if source is not None:
            self._node_source = source
        if source == "line_token_ids":
            z = np.load(lt)
            row = {int(l): k for k, l in enumerate(z["lineno"].tolist())}
            L = self.line_len
            ids = torch.ones((len(lineno), L), dtype=torch.int64)                               # pad id 1; a line without ids = one <s>
            ids[:, 0] = 0
            zi = z["ids"].astype(np.int64)
            w = min(L, zi.shape[1])                                                             # pad (id 1) or truncate to FUSED.LINE_LEN
            for k, l in enumerate(lineno):
                if l in row:
                    ids[k, :w] = torch.from_numpy(zi[row[l], :w])
                    ids[k, w:] = 1
            g.ndata["_token_ids"] = ids
        elif source == "node_emb":
            z = np.load(ne)
            row = {int(l): k for k, l in enumerate(z["lineno"].tolist())}
            emb = torch.zeros((len(lineno), z["emb"].shape[1]), dtype=torch.float32)
            for k, l in enumerate(lineno):
                if l in row:
                    emb[k] = torch.from_numpy(z["emb"][row[l]].astype(np.float32))
            g.ndata["_UNIX_NODE_EMB"] = emb
        else:
            raise FileNotFoundError(f"{_id}: neither line_token_ids/{_id}.npz nor node_emb/{_id}.npz under {self.root}")
        return g

    def __getitem__(self, i):
        import os
        import numpy as np
        path, label = self.items[i]
        _id = int(os.path.basename(path).rsplit(".png", 1)[0])                                  # data_list.py:123
        g = self.graph(_id)
        if self.fused:
            from PIL import Image
            with Image.open(self._path(path)) as im:                                           # pil_loader :49-52
                img = torch.from_numpy(np.asarray(im.convert("RGB")).copy())
            ids = torch.from_numpy(np.load(self._path("token_ids", f"{_id}.npy")).astype(np.int64))[: self.seq_len]
            if ids.numel() < self.seq_len:
                ids = torch.cat([ids, torch.ones(self.seq_len - ids.numel(), dtype=torch.int64)])
            return g, img, ids, label
        if self._func_emb is None:
            import pandas as pd
            df = pd.read_pickle(self._path("result.pkl"))                                       # :101-103
            self._func_emb = {int(k): torch.tensor(v, dtype=torch.float32) for k, v in zip(df.ids.tolist(), df.repr.tolist())}
        img_emb = torch.load(self._path("swinv2_method_level_try5", f"{_id}.pt"), map_location="cpu").float().view(-1)     # :133
        return g, img_emb, self._func_emb[_id], label


def collate(samples):
    gs, a, b, y = zip(*samples)
    g = batch_graphs(list(gs))
    # the CSR index is built on the DEVICE once the edge lists are there (model_step_inputs -> BatchedGraph.index: 0.1 ms instead of
    # ~20 ms of host sorting per batch of 32 graphs)
    # decoded images may arrive as uint8 [H, W, 3] of any size (device-side resize + normalise: data/image_ingest.py): kept as a list
    imgs = list(a) if a[0].dtype == torch.uint8 else torch.stack(a)
    return g, imgs, torch.stack(b), torch.tensor(y, dtype=torch.int64)


def _world():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(), dist.get_rank()
    return 1, 0


def bigvul_dataset(config):
    f = config.FUSED
    fused = bool(f.ENABLE)
    root = getattr(f, "DATA_ROOT", "")
    if root:
        return tuple(BigVulFiles(root, split, config, fused) for split in ("train", "val", "test"))
    train = SyntheticBigVul(f.SYNTH_TRAIN, 0, config, fused)
    val = SyntheticBigVul(f.SYNTH_VAL, 10_000_000, config, fused)
    test = SyntheticBigVul(f.SYNTH_TEST, 20_000_000, config, fused)
    return train, val, test


def bigvul_loader_graph(config):
    train_data, val_data, test_data = bigvul_dataset(config)
    world, rank = _world()
    s_train = DistributedSampler(train_data, num_replicas=world, rank=rank, shuffle=True, drop_last=True)
    s_val = DistributedSampler(val_data, num_replicas=world, rank=rank, shuffle=config.TEST.SHUFFLE)
    s_test = DistributedSampler(test_data, num_replicas=world, rank=rank, shuffle=config.TEST.SHUFFLE)
    kw = dict(batch_size=config.DATA.BATCH_SIZE, num_workers=0, pin_memory=config.DATA.PIN_MEMORY, collate_fn=collate)
    loader_train = DataLoader(train_data, sampler=s_train, drop_last=True, **kw)
    loader_val = DataLoader(val_data, sampler=s_val, drop_last=False, **kw)
    loader_test = DataLoader(test_data, sampler=s_test, drop_last=False, **kw)
    return train_data, val_data, test_data, loader_train, loader_val, loader_test, None
