"""``bigvul_loader_graph(config)`` with the reference's return signature (mvuld/data/bigvul_dataset.py:157-216):
``train_data, val_data, test_data, loader_train, loader_val, loader_test, mixup_fn``.

The Big-Vul corpus, the Joern graphs, the rendered PNGs and the cached encoder features are not on this box (dataset is
an external download, README.md:32), so samples come from ``data.synthetic`` with the schema the reference's
``ImageList.__getitem__`` yields (data_list.py:107-141).  Sharding is the reference's: ``DistributedSampler`` per split,
``shuffle=True, drop_last=True`` for train (bigvul_dataset.py:163-185), graphs collated with ``graph.batch`` (dgl.batch).

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
