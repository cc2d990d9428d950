"""Data parallelism for the fused step: one process per GPU, RCCL (torch.distributed backend "nccl" on ROCm) over
xGMI, replacing ``torch.nn.parallel.DistributedDataParallel(broadcast_buffers=False, find_unused_parameters=True)``
of main_bigvul.py:162-164.

* parameters are broadcast once from rank 0 (DDP construction); BatchNorm running stats stay per rank
  (``broadcast_buffers=False``, no SyncBN);
* gradients live in ONE flat fp32 buffer (optimizer.ParamStore), so the per-step exchange is a handful of large
  all-reduces on contiguous ranges instead of 25 MB DDP buckets: xGMI is point-to-point, ring collectives are
  per-link bound, so fewer/larger messages amortise the per-collective latency.  Each encoder's range is launched
  (async) from the backward of its FIRST block -- the point where all its gradients are final -- so the exchange
  overlaps the rest of backward; whatever is left goes out after backward;
* unused parameters simply have zero gradients in the flat buffer (no find_unused_parameters machinery);
* ``gloo`` on CPU for the multi-process logic tests.
"""
import os

import torch
import torch.distributed as dist


def init_distributed(local_rank=None, backend=None):
    """Initialise from the launcher's env (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*).  Returns (rank, world, local)."""
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", local_rank if local_rank is not None else 0))
    if world > 1 and not dist.is_initialized():
        use_cuda = torch.cuda.is_available()
        backend = backend or ("nccl" if use_cuda else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if use_cuda:
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, init_method="env://", world_size=world, rank=rank)
    return rank, world, local


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def get_rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def barrier():
    if world_size() > 1:
        dist.barrier()


def broadcast_parameters(flat: torch.Tensor, src=0):
    if world_size() > 1:
        dist.broadcast(flat, src)


class GradAllReducer:
    """Average the flat gradient buffer across ranks in a few large async all-reduces."""

    def __init__(self, flat_grad: torch.Tensor, max_bucket_elems=64 * 1024 * 1024):
        self.g = flat_grad
        self.max_bucket = max_bucket_elems
        self.pending = []
        self.done = []            # [(start, end)] already launched this step

    def _launch(self, a, b):
        while a < b:
            e = min(b, a + self.max_bucket)
            seg = self.g[a:e]
            self.pending.append((dist.all_reduce(seg, op=dist.ReduceOp.SUM, async_op=True), a, e))
            a = e

    def launch_ranges(self, ranges):
        """Start the exchange of finished ranges (called from inside backward)."""
        if world_size() == 1:
            return
        for a, b in ranges:
            self._launch(a, b)
            self.done.append((a, b))

    def finish(self):
        """Exchange everything not yet launched and wait.  The buffer then holds SUMS over ranks: the 1/world factor
        is folded into the clip coefficient the AdamW kernel applies (ParamStore.clip_grad_norm_(grad_scale=1/world))."""
        ws = world_size()
        if ws == 1:
            return
        if self.g.is_cuda:
            from . import ops
            ops.join_wgrad_stream()           # gradients still being accumulated on the weight-gradient stream
        covered = sorted(self.done)
        pos = 0
        for a, b in covered + [(self.g.numel(), self.g.numel())]:
            if a > pos:
                self._launch(pos, a)
            pos = max(pos, b)
        for w, _, _ in self.pending:
            w.wait()
        self.pending.clear()
        self.done.clear()
