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


def gather_cat(t: torch.Tensor) -> torch.Tensor:
    """Concatenation over ranks of equally shaped tensors (validation outputs: every rank then derives the same metrics)."""
    if world_size() == 1:
        return t
    parts = [torch.empty_like(t) for _ in range(world_size())]
    dist.all_gather(parts, t.contiguous())
    return torch.cat(parts, 0)


def broadcast_parameters(flat: torch.Tensor, src=0):
    if world_size() > 1 or (os.environ.get("MVULD_FORCE_ALLREDUCE", "0") == "1" and dist.is_available() and dist.is_initialized()):
        dist.broadcast(flat, src)


def attach_gradient_exchange(store, max_bucket_elems=64 * 1024 * 1024, payload=None):
    """The data-parallel wiring of one fused training step (what DDP's reducer hooks are to the reference, main_bigvul.py:162-164),
    shared by bench.py and main_bigvul.py: a GradAllReducer over the store's flat gradient buffer whose per-range launches fire from
    inside backward -- each Swin stage when its first block has launched its last backward kernel (stage 3 and 2 hold 95 % of the
    Swin gradients and finish early), the text encoder from its first op; everything else goes out with `reducer.finish()`.
    Also sets the 1/world factor the clip coefficient folds in.  Returns the reducer (call .finish() after backward)."""
    from . import ops
    reducer = GradAllReducer(store.grad, max_bucket_elems, payload=payload)
    store.grad_scale = 1.0 / world_size()
    if reducer._active() and store.grad.is_cuda and "MVULD_GEMM_DYNAMIC_TILES" not in os.environ:
        # the collectives' channels stay resident on some CUs for milliseconds while backward goes on beside them: the persistent NT grid then
        # claims ALL its tiles at run time (include/mvuld_hip.h: mvuld_set_gemm_dynamic_tiles; with 8-64 CUs held, tools/microbench/
        # cu_hog_gemm.hip measures the K = 512 products 13-24 % shorter than under the static walk, profiles/r04_dynamic_tiles.txt)
        from . import hip
        hip.LIB.fn("mvuld_set_gemm_dynamic_tiles")(2)
    tags = [f"swin.layers.{i}" for i in range(4)] + ["unixcoder"]
    for tag in tags:
        if reducer._active():
            ops.on_backward_done(tag, lambda tag=tag: reducer.launch_ranges(store.segment(tag + ".")), key="grad-exchange")
        else:
            ops.on_backward_done(tag, None, key="grad-exchange")
    return reducer


class GradAllReducer:
    """Average the flat gradient buffer across ranks in a few large async all-reduces.

    payload: "fp32" (default) sums the fp32 buffer in place -- 0.93 GB per step and GPU for the 232 M parameters.  "bf16" (env
    MVULD_GRAD_EXCHANGE=bf16, north_star's 464 MB payload) sends each range rounded to bf16: the range is cast into a staging buffer
    (the library's cast kernel on the GPU), the staging buffer is all-reduced -- the SUM over ranks is then formed in bf16 by the
    collective --, and finish() widens it back into the fp32 buffer.  Half the wire bytes for one bf16 rounding of every rank's
    contribution plus the collective's bf16 additions (|error| <= ~world * 2^-8 of the sum of the ranks' magnitudes);
    the fp32 exchange stays the default until a measured 8-GPU step shows the wire time on the critical path."""

    def __init__(self, flat_grad: torch.Tensor, max_bucket_elems=64 * 1024 * 1024, payload=None):
        self.g = flat_grad
        self.max_bucket = max_bucket_elems
        self.payload = (payload or os.environ.get("MVULD_GRAD_EXCHANGE", "fp32")).lower()
        if self.payload not in ("fp32", "bf16"):
            raise ValueError(f"gradient exchange payload {self.payload!r}: fp32 | bf16")
        self.stage = None         # bf16 staging buffer of the whole gradient, allocated on first use
        self.pending = []
        self.done = []            # [(start, end)] already launched this step
        # a single-rank process group still runs every collective when forced: lets a one-GPU box execute the RCCL path end to end
        self.force = os.environ.get("MVULD_FORCE_ALLREDUCE", "0") == "1"

    def _active(self):
        return world_size() > 1 or (self.force and dist.is_available() and dist.is_initialized())

    def _launch(self, a, b):
        while a < b:
            e = min(b, a + self.max_bucket)
            seg = self.g[a:e]
            if self.payload == "bf16":
                if self.stage is None:
                    self.stage = torch.empty(self.g.numel(), dtype=torch.bfloat16, device=self.g.device)
                st = self.stage[a:e]
                self._cast(seg, st)
                self.pending.append((dist.all_reduce(st, op=dist.ReduceOp.SUM, async_op=True), a, e))
            else:
                self.pending.append((dist.all_reduce(seg, op=dist.ReduceOp.SUM, async_op=True), a, e))
            a = e

    @staticmethod
    def _cast(src, dst):
        """dst = src in dst's dtype: the library's cast kernel on the GPU (stream-ordered with the collective that follows on the
        same stream), torch on the CPU (gloo tests: no kernel runs there)."""
        if src.is_cuda:
            from .hip import BF16, F32, call, ptr
            code = {torch.float32: F32, torch.bfloat16: BF16}
            call("cast", ptr(src), code[src.dtype], ptr(dst), code[dst.dtype], src.numel())
        else:
            dst.copy_(src)

    def launch_ranges(self, ranges):
        """Start the exchange of finished ranges (called from inside backward)."""
        if not self._active():
            return
        for a, b in ranges:
            self._launch(a, b)
            self.done.append((a, b))

    def finish(self):
        """Exchange everything not yet launched and wait.  The buffer then holds SUMS over ranks: the 1/world factor
        is folded into the clip coefficient the AdamW kernel applies (ParamStore.clip_grad_norm_(grad_scale=1/world))."""
        if not self._active():
            return
        if self.g.is_cuda:
            from . import ops
            ops.join_wgrad_stream()           # gradients still being accumulated on the weight-gradient stream
            ops.join_grad_streams()           # ... or on any other stream of the step, whatever order autograd replayed them in
        covered = sorted(self.done)
        pos = 0
        for a, b in covered + [(self.g.numel(), self.g.numel())]:
            if a > pos:
                self._launch(pos, a)
            pos = max(pos, b)
        for w, a, e in self.pending:
            w.wait()
            if self.payload == "bf16":
                self._cast(self.stage[a:e], self.g[a:e])
        self.pending.clear()
        self.done.clear()
