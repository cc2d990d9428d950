"""The fused multimodal model the north_star names: SwinV2 image encoder + UniXcoder text encoder + graph/fusion
head in ONE forward/backward (the reference runs the encoders offline and caches their outputs:
mvuld/data/data_list.py:179-211,265-317).  Composition of the reference module
boundaries: ``img_embedding = swin.forward_features(images)``, ``func_text_embedding = unixcoder.get_xcode_vec(ids)[1]``,
``logits = head(g, img_embedding, func_text_embedding)``.  State-dict prefixes: ``swin.``, ``unixcoder.``, ``head.``."""
import os

import torch
import torch.nn as nn

from .GraphModel import Multi_DefectModel_new_GCN, cross_entropy
from .swin_transformer_v2 import SwinTransformerV2
from .unixcoder import MyUniXcoder, RobertaConfigLite, RobertaModel


# Stream priorities of the side streams (experiment knob MVULD_STREAM_PRIO="text,wgrad,graph", e.g. "1,1,1" = all below the image encoder's
# stream; HIP: larger number = lower priority, clamped to the device's range).  Default: all equal.
_PRIO = dict(zip(("text", "wgrad", "graph"), (int(v) for v in os.environ.get("MVULD_STREAM_PRIO", "0,0,0").split(","))))

class FusedMVulD(nn.Module):
    def __init__(self, config, roberta_config=None, act_dtype=torch.bfloat16, swin=None):
        super().__init__()
        from .build import build_model
        self.act_dtype = act_dtype
        self.swin = swin if swin is not None else build_model(config, act_dtype)
        rc = roberta_config or RobertaConfigLite()
        self.unixcoder = MyUniXcoder(RobertaModel(rc, act_dtype), rc)
        for p in self.unixcoder.classifier.parameters():      # not on the fused path
            p.requires_grad_(False)
        for p in self.swin.head.parameters():
            p.requires_grad_(False)
        from .GraphModel import head_class
        head_name = str(getattr(getattr(config, "FUSED", None), "HEAD", "") or "Multi_DefectModel_new_GCN")
        self.head = head_class(head_name)(config, act_dtype=act_dtype)
        self._split_head = hasattr(self.head, "forward_graph")     # the full head's graph branch can run beside the encoders
        self._side = None
        self._gs = None
        self._tn_budget = -1
        self._wg = None
        self._inflight = []
        self.max_steps_in_flight = 2          # 0 = do not throttle the host (bench.py's enqueue-cost measurement)
        for n, p in self.head.named_parameters():
            if n.startswith(tuple(self.head.unused_parameter_prefixes)):
                p.requires_grad_(False)

    def no_weight_decay(self):
        return {"swin." + k for k in self.swin.no_weight_decay()}

    def no_weight_decay_keywords(self):
        return self.swin.no_weight_decay_keywords()

    def forward(self, g, images, source_ids, seq_lens=None, node_ids=None, node_lens=None):
        from .. import ops
        if not (ops.FP8_FWD[0] and images.is_cuda):
            return self._forward(g, images, source_ids, seq_lens, node_ids, node_lens)
        # fp8 forward GEMMs (configs[4]): last pass's amax of every fused quantisation site becomes this pass's scale, once for both
        # encoders and before their streams fork
        ops.fp8_roll(images.device)
        ops.FP8_IN_FUSED[0] = True
        try:
            return self._forward(g, images, source_ids, seq_lens, node_ids, node_lens)
        finally:
            ops.FP8_IN_FUSED[0] = False

    def _forward(self, g, images, source_ids, seq_lens=None, node_ids=None, node_lens=None):
        """The two encoders are independent until the head: the text encoder runs on a second HIP stream so its kernels fill
        the tails of the image encoder's launches (and vice versa); autograd replays each branch's backward on the stream its
        forward ran on.  The side stream is joined before the head and, in backward, when the text encoder's first op has
        launched its last kernel (its parameter gradients are atomics into the flat store, invisible to autograd's own
        stream bookkeeping).  MVULD_CONCURRENT=0, or the per-launch timing mode, keeps everything on one stream.
        seq_lens (optional, host int tensor [B]): non-pad tokens per function; with it the text encoder runs pad-free on the packed
        tokens (same sentence vectors: pad rows never reach them, unixcoder.py:35-37); MVULD_PACK_TEXT=0 ignores it.
        node_ids / node_lens (optional): token ids [sum N, L] (+ host-side non-pad counts) of the source line behind every graph node;
        the node features `_UNIX_NODE_EMB` are then computed here, on the device, by the same text encoder (no gradient, as the
        reference caches them offline: data_list.py:265-317) instead of being read from g.ndata."""
        from .. import hip, ops
        concurrent = images.is_cuda and os.environ.get("MVULD_CONCURRENT", "1") != "0" and not hip.TIMING.enabled
        if images.is_cuda and self.training:
            # weight-gradient kernels beside the chain: planned for half the chip (gemm_tn256.hip); alone (one stream): all of it
            if self._tn_budget < 0:
                self._tn_budget = torch.cuda.get_device_properties(images.device).multi_processor_count // 2
            hip.LIB.fn("mvuld_set_gemm_tn256_budget")(self._tn_budget if concurrent else 0)
        if os.environ.get("MVULD_PACK_TEXT", "1") == "0":
            seq_lens = None
        self.unixcoder.return_tokens = False                           # only the sentence vector is read here
        if node_ids is not None:
            was = self.unixcoder.training
            self.unixcoder.eval()                                      # offline feature extraction semantics: no dropout, no gradient
            g.ndata["_UNIX_NODE_EMB"] = self.unixcoder.encode_lines(node_ids, node_lens).float()
            self.unixcoder.train(was)
        if torch.is_grad_enabled() and self.training:
            ops.begin_step()
        if not concurrent:
            ops.WGRAD_STREAM[0] = None
            ops.on_backward_done("unixcoder", None, key="fused-join")
            img = self.swin.forward_features(images)                   # [B,1024]
            _, txt = self.unixcoder.get_xcode_vec(source_ids, seq_lens)  # [B,768]
            hfeat = self.head.forward_graph(g) if self._split_head else None
        else:
            main = torch.cuda.current_stream(images.device)
            # Bound how far the host may run ahead of the GPU: tensors handed to another stream (record_stream) cannot be reused
            # until that stream has passed them, so an unthrottled host (24 ms of enqueue per 66 ms step) keeps a few hundred MB
            # more alive for every step it is ahead.  Two steps in flight lose nothing.
            if not torch.cuda.is_current_stream_capturing() and self.max_steps_in_flight > 0:
                ev = torch.cuda.Event()
                ev.record(main)
                self._inflight.append(ev)
                while len(self._inflight) > self.max_steps_in_flight:
                    self._inflight.pop(0).synchronize()
            if self._side is None:
                self._side = torch.cuda.Stream(device=images.device, priority=_PRIO["text"])
            side = self._side
            if self._wg is None:
                self._wg = torch.cuda.Stream(device=images.device, priority=_PRIO["wgrad"])
            ops.register_grad_stream(side)
            ops.register_grad_stream(self._wg)
            # third stream: the image encoder's weight gradients (nothing in backward depends on them); joined into the main
            # stream when the encoder's first op has finished its backward (ops.fire_backward_done("swin"))
            use_wg = torch.is_grad_enabled() and self.training and os.environ.get("MVULD_WGRAD_STREAM", "1") != "0"
            ops.WGRAD_STREAM[0] = (main.cuda_stream, self._wg) if use_wg else None
            side.wait_stream(main)
            gst = side
            if self._split_head and os.environ.get("MVULD_GRAPH_STREAM", "1") != "0":
                if self._gs is None:
                    self._gs = torch.cuda.Stream(device=images.device, priority=_PRIO["graph"])
                gst = self._gs
                ops.register_grad_stream(gst)
                gst.wait_stream(main)                                  # inputs are ready; nothing of this step is on `main` yet
            # Side stream: the text encoder and the head's graph branch (which needs neither encoder: many small launches
            # that leave most of the chip idle), both under the image encoder's dense kernels on the main stream.  Host
            # order text -> image -> graph: autograd replays later-created nodes first, so in backward the graph branch is
            # enqueued at once, the image encoder next and the text encoder last -- whose first op, the last side-stream
            # work of the step, fires the join below.
            with torch.cuda.stream(side):
                _, txt = self.unixcoder.get_xcode_vec(source_ids, seq_lens)
            img = self.swin.forward_features(images)
            # The graph branch gets a stream of its own: ~600 small, latency-bound launches per step that used to queue in front of
            # (forward) and behind (backward: 6 ms) the text encoder on the side stream, which had become the last to finish.
            hfeat = None
            if self._split_head:
                with torch.cuda.stream(gst):
                    hfeat = self.head.forward_graph(g)
            main.wait_stream(side)
            if gst is not side:
                main.wait_stream(gst)
            txt.record_stream(main)
            if hfeat is not None:
                hfeat.record_stream(main)

            def join():                                                # runs inside backward, on the side stream
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(images.device))
                main.wait_event(ev)
            ops.on_backward_done("unixcoder", join, key="fused-join")
        g.ndata["_FUNC_EMB"] = txt                                     # (per-node repeat is dead in the reference head)
        if not self._split_head:                                       # an ablation head (FUSED.HEAD): whole head after the join
            return self.head(g, img, txt)
        return self.head.forward_join(g, img, txt, hfeat)
