"""A whole fused training step -- forward, cross-entropy, backward on all three streams, gradient-norm clip, fused AdamW, the
refresh of the transposed weight copies -- captured ONCE in a hipGraph and replayed: one graph launch per step instead of ~1900
kernel launches through Python / ctypes (about 30 ms of host time per step at batch 32).

What changes from step to step lives in device memory the captured kernels read, not in their (frozen) arguments:
  * the learning rate of the per-iteration cosine schedule and Adam's bias corrections: `FusedAdamW.dev_hyper`, refreshed by a
    12-byte-per-group asynchronous copy from pinned memory before each replay;
  * the random masks (DropPath, head / text-encoder dropouts, attention-probability dropout): every RNG kernel mixes a device-resident
    step counter (`ops.RNG_OFFSET`) into its host seed; the counter is advanced by a kernel inside the graph;
  * the batch: static input tensors (`copy_` the next batch into them).  Shapes must not change between replays: fixed batch size
    and image size, token ids padded to the model's sequence length -- and, with the pad-free text encoder, a fixed PACKING PLAN
    (total non-pad tokens), i.e. the synthetic benchmark batch, or real batches padded to a token budget.
Single-GPU only here: with N > 1 ranks the step runs eagerly (the RCCL exchange is launched from inside backward by Python hooks).
"""
import torch

from . import ops
from .hip import call, ptr


class GraphedTrainStep:
    def __init__(self, model, optimizer, scheduler, loss_fn, inputs, labels, clip_grad, model_kwargs=None, warmup=3):
        """inputs: tuple of static positional inputs of `model`; labels: static target tensor; loss_fn(logits, labels) -> (loss, _)."""
        assert torch.cuda.is_available()
        self.model, self.opt, self.sched, self.loss_fn = model, optimizer, scheduler, loss_fn
        self.inputs, self.labels, self.kw, self.clip = inputs, labels, dict(model_kwargs or {}), clip_grad
        dev = labels.device
        self.it = 0
        self.counter = torch.zeros(1, dtype=torch.int64, device=dev)
        ops.RNG_OFFSET[0] = self.counter
        ng = len(optimizer.param_groups)
        optimizer.dev_hyper = torch.zeros((ng, 3), dtype=torch.float32, device=dev)
        # staging ring for {lr, bias corrections}: a slot is rewritten only after the copy that last read it has run (its event);
        # a replay costs microseconds on the host and tens of ms on the GPU, so nothing else throttles an unsynchronised step() loop
        self._pinned = [torch.empty((ng, 3), dtype=torch.float32).pin_memory() for _ in range(4)]
        self._pinned_ev = [None] * len(self._pinned)
        if hasattr(model, "max_steps_in_flight"):
            model.max_steps_in_flight = 0
        # warm-up on a side stream: allocator pools, workspaces, transposed-weight job tables, LDS attribute calls ... all exist
        # before the capture
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(warmup):
                self._upload_hyper()
                self._eager_body()
                self.it += 1
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        self._upload_hyper()
        step_before = self.opt._step
        with torch.cuda.graph(self.graph):
            self.loss, self.norm = self._eager_body()
        self.opt._step = step_before     # the capture pass ran the optimizer's Python bookkeeping but none of its kernels
        self.replays = 0

    def _eager_body(self):
        call("counter_add", ptr(self.counter), 1)
        logits = self.model(*self.inputs, **self.kw)
        loss, _ = self.loss_fn(logits, self.labels)
        loss.backward()
        norm = self.opt.clip_grad_norm_(self.clip)
        self.opt.step()
        self.opt.zero_grad()
        return loss, norm

    def _upload_hyper(self):
        self.sched.step_update(self.it)                          # sets param_group["lr"] for this iteration (cosine, per iteration)
        slot = self.it % len(self._pinned)
        buf = self._pinned[slot]
        if self._pinned_ev[slot] is not None:
            self._pinned_ev[slot].synchronize()                   # the H2D copy queued len(ring) steps ago has consumed this slot
        buf.copy_(self.opt.host_hyper())
        self.opt.dev_hyper.copy_(buf, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.labels.device))
        self._pinned_ev[slot] = ev

    def step(self):
        """One training step = one graph launch.  Returns the (device) loss and gradient-norm tensors of that step."""
        self._upload_hyper()
        self.opt._step += 1                                      # FusedAdamW.step() inside the graph does not run its Python again
        self.graph.replay()
        self.it += 1
        self.replays += 1
        return self.loss, self.norm

    def close(self):
        ops.RNG_OFFSET[0] = None
        self.opt.dev_hyper = None
        if hasattr(self.model, "max_steps_in_flight"):
            self.model.max_steps_in_flight = 2
