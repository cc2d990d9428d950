"""UniXcoder text encoder on libmvuld_hip.so.

Mirrors the reference surface of ``mvuld/models/unixcoder.py``: ``MyUniXcoder(encoder, config, tokenizer,
tokenize)`` with ``get_xcode_vec`` (:33-38), ``forward`` (:40-54), ``get_repr`` (:91-95), ``myEncode`` (:56-68) and
``UniXcoder(model_name)`` with ``.model/.config/.tokenizer/.tokenize`` (:97-152) -- but the encoder is this
file's own RoBERTa-shaped module (state_dict keys of HF ``RobertaModel``: ``embeddings.*``,
``encoder.layer.N.attention.self.{query,key,value}``, ``attention.output``, ``intermediate``, ``output``,
``pooler``) executed by hand-written kernels:

* q/k/v projections are one fused ``[3H, H]`` GEMM (the three weights are stored adjacently; the
  state_dict still exposes ``query/key/value`` separately);
* attention is the fused pad-masked kernel: the reference's 3-D mask ``mask[:,None,:]*mask[:,:,None]``
  (:35-36; additive -10000 in transformers 4.18) is evaluated from the ``valid`` vector, no ``[B,L,L]`` tensor;
* post-LN blocks are LayerNorm(dense + input) in one kernel; GELU(erf) rides the GEMM epilogue.

Generation (``generate`` / ``Beam``, :176-342) is not on MVulD's path and is not provided.
"""
import math

import torch
import torch.nn as nn

from .. import hip, ops
from ..hip import call, ptr, dt


class RobertaConfigLite:
    """The fields of transformers.RobertaConfig this path reads (defaults: microsoft/unixcoder-base-nine shape)."""

    def __init__(self, vocab_size=51416, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                 intermediate_size=3072, max_position_embeddings=1026, type_vocab_size=10, pad_token_id=1,
                 layer_norm_eps=1e-5, hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1, **kw):
        self.vocab_size, self.hidden_size = vocab_size, hidden_size
        self.num_hidden_layers, self.num_attention_heads = num_hidden_layers, num_attention_heads
        self.intermediate_size, self.max_position_embeddings = intermediate_size, max_position_embeddings
        self.type_vocab_size, self.pad_token_id, self.layer_norm_eps = type_vocab_size, pad_token_id, layer_norm_eps
        self.hidden_dropout_prob, self.attention_probs_dropout_prob = hidden_dropout_prob, attention_probs_dropout_prob
        self.is_decoder = True            # set by the reference (unixcoder.py:109); no effect on a 3-D mask
        self.eos_token_id = 2


# ------------------------------------------------------------------------------------------------ functions
class _EmbedFn(torch.autograd.Function):
    """word + position + token-type(0) embeddings, LayerNorm (HF RobertaEmbeddings)."""

    @staticmethod
    def forward(ctx, word_w, ids, emb, act_dtype):
        hip.require_gpu(ids)
        B, L = ids.shape
        cfg = emb.config
        H = cfg.hidden_size
        ids = ids.contiguous()
        pos = torch.empty((B, L), dtype=torch.int32, device=ids.device)
        valid = torch.empty((B, L), dtype=torch.int32, device=ids.device)
        call("position_ids", ptr(ids), ptr(pos), ptr(valid), B, L, cfg.pad_token_id)
        raw = torch.empty((B * L, H), dtype=act_dtype, device=ids.device)
        call("embed_fwd", ptr(ids), ptr(pos), ptr(emb.word_embeddings.weight), ptr(emb.position_embeddings.weight),
             ptr(emb.token_type_embeddings.weight), ptr(raw), B * L, H, cfg.vocab_size, cfg.max_position_embeddings, dt(raw))
        y, mean, rstd, _ = ops.layernorm_fwd(raw, emb.LayerNorm.weight.data, emb.LayerNorm.bias.data, cfg.layer_norm_eps)
        ctx.save_for_backward(ids, pos, raw, mean, rstd)
        ctx.emb = emb
        ctx.mark_non_differentiable(valid)
        return y, valid

    @staticmethod
    def backward(ctx, dy, _dvalid):
        ids, pos, raw, mean, rstd = ctx.saved_tensors
        emb = ctx.emb
        cfg = emb.config
        H = cfg.hidden_size
        draw = ops.layernorm_bwd(dy.contiguous(), raw, emb.LayerNorm.weight, emb.LayerNorm.bias, mean, rstd)
        call("embed_bwd", ptr(ids), ptr(pos), ptr(draw), ptr(ops.grad_of(emb.word_embeddings.weight)),
             ptr(ops.grad_of(emb.position_embeddings.weight)), ids.numel(), H, cfg.vocab_size, cfg.max_position_embeddings, dt(draw))
        ops.colsum_into(draw, ops.grad_of(emb.token_type_embeddings.weight)[0])
        ops.fire_backward_done("unixcoder")     # first op of the encoder: every UniXcoder gradient is final now
        return None, None, None, None


_SEED = [0xC0DE5EED]


def _next_seed():
    _SEED[0] = (_SEED[0] * 6364136223846793005 + 1442695040888963407) & 0xFFFFFFFFFFFFFFFF
    return _SEED[0]


class _DropoutFn(torch.autograd.Function):
    """Hidden-state dropout (HF hidden_dropout_prob) with a counter-based mask: backward replays the same (seed, index) hash."""

    @staticmethod
    def forward(ctx, x, p, seed):
        ctx.meta = (p, seed)
        return ops.dropout(x, p, seed)

    @staticmethod
    def backward(ctx, dy):
        p, seed = ctx.meta
        return ops.dropout(dy.contiguous(), p, seed), None, None


class _PackEmbedFn(torch.autograd.Function):
    """Packed variant of _EmbedFn: only the non-pad tokens are embedded; row cu[b] + r of the result is the r-th non-pad token of
    sequence b (the reference computes the pad rows too and never uses them: unixcoder.py:35-37)."""

    @staticmethod
    def forward(ctx, word_w, ids, cu, T, emb, act_dtype):
        hip.require_gpu(ids)
        B, L = ids.shape
        cfg = emb.config
        H = cfg.hidden_size
        ids = ids.contiguous()
        ids_p = torch.empty((T,), dtype=torch.int64, device=ids.device)
        pos_p = torch.empty((T,), dtype=torch.int32, device=ids.device)
        rowmap = torch.empty((T,), dtype=torch.int32, device=ids.device)
        call("pack_tokens", ptr(ids), ptr(cu), ptr(ids_p), ptr(pos_p), ptr(rowmap), B, L, cfg.pad_token_id)
        raw = torch.empty((T, H), dtype=act_dtype, device=ids.device)
        call("embed_fwd", ptr(ids_p), ptr(pos_p), ptr(emb.word_embeddings.weight), ptr(emb.position_embeddings.weight),
             ptr(emb.token_type_embeddings.weight), ptr(raw), T, H, cfg.vocab_size, cfg.max_position_embeddings, dt(raw))
        y, mean, rstd, _ = ops.layernorm_fwd(raw, emb.LayerNorm.weight.data, emb.LayerNorm.bias.data, cfg.layer_norm_eps)
        ctx.save_for_backward(ids_p, pos_p, raw, mean, rstd)
        ctx.emb = emb
        ctx.mark_non_differentiable(rowmap)
        return y, rowmap

    @staticmethod
    def backward(ctx, dy, _drow):
        ids_p, pos_p, raw, mean, rstd = ctx.saved_tensors
        emb = ctx.emb
        cfg = emb.config
        H = cfg.hidden_size
        draw = ops.layernorm_bwd(dy.contiguous(), raw, emb.LayerNorm.weight, emb.LayerNorm.bias, mean, rstd)
        call("embed_bwd", ptr(ids_p), ptr(pos_p), ptr(draw), ptr(ops.grad_of(emb.word_embeddings.weight)),
             ptr(ops.grad_of(emb.position_embeddings.weight)), ids_p.numel(), H, cfg.vocab_size, cfg.max_position_embeddings, dt(draw))
        ops.colsum_into(draw, ops.grad_of(emb.token_type_embeddings.weight)[0])
        ops.fire_backward_done("unixcoder")
        return None, None, None, None, None, None


class _SegmentMeanFn(torch.autograd.Function):
    """Sentence vector over packed tokens: mean of rows cu[b] .. cu[b+1]-1  (= (tok * mask).sum(1) / mask.sum(-1), unixcoder.py:37)."""

    @staticmethod
    def forward(ctx, tok, cu, B):
        H = tok.shape[1]
        out = torch.empty((B, H), dtype=tok.dtype, device=tok.device)
        call("segment_mean_fwd", ptr(tok), ptr(cu), ptr(out), B, H, dt(tok))
        ctx.save_for_backward(cu)
        ctx.dims = (tok.shape[0], B, H)
        return out

    @staticmethod
    def backward(ctx, dout):
        (cu,) = ctx.saved_tensors
        T, B, H = ctx.dims
        dx = torch.empty((T, H), dtype=dout.dtype, device=dout.device)
        call("segment_mean_bwd", ptr(dout.contiguous()), ptr(cu), ptr(dx), B, H, dt(dx))
        return dx, None, None


class _UnpackFn(torch.autograd.Function):
    """Packed rows -> the reference's padded [B*L, H] layout (pad rows zero; the reference leaves unused values there)."""

    @staticmethod
    def forward(ctx, tok, rowmap, rows):
        out = torch.zeros((rows, tok.shape[1]), dtype=tok.dtype, device=tok.device)
        call("rows_map", ptr(tok), ptr(rowmap), ptr(out), tok.shape[0], tok.shape[1], 1, dt(tok))
        ctx.save_for_backward(rowmap)
        return out

    @staticmethod
    def backward(ctx, dout):
        (rowmap,) = ctx.saved_tensors
        dout = dout.contiguous()
        dx = torch.empty((rowmap.shape[0], dout.shape[1]), dtype=dout.dtype, device=dout.device)
        call("rows_map", ptr(dout), ptr(rowmap), ptr(dx), rowmap.shape[0], dout.shape[1], 0, dt(dout))
        return dx, None, None


class _PinnedRing:
    """Small pinned host buffers handed out round-robin: a tiny tensor built on the host every step (cu_seqlens) reaches the device
    by an ASYNCHRONOUS copy.  (From pageable memory `.to(device)` synchronises the stream: the host would then run in lock-step
    with the GPU instead of a step or two ahead of it.)  A slot is reused 16 copies later, long after its copy has executed."""

    def __init__(self, slots=16, elems=4096):
        self.bufs, self.i, self.elems, self.slots = None, 0, elems, slots

    def stage(self, t: torch.Tensor, device):
        if not torch.cuda.is_available() or device.type != "cuda" or t.numel() > self.elems or t.dtype != torch.int32:
            return t.to(device)
        if self.bufs is None:
            self.bufs = [torch.empty(self.elems, dtype=torch.int32).pin_memory() for _ in range(self.slots)]
        b = self.bufs[self.i % self.slots][:t.numel()]
        self.i += 1
        b.copy_(t)
        return b.to(device, non_blocking=True)


_CU_STAGE = _PinnedRing()


class PackedSeqs:
    """cu_seqlens of a packed batch: `cu` int32 [B+1] on the device, total tokens, and the sum of squared lengths."""

    def __init__(self, cu, total, sumsq):
        self.cu, self.total, self.sumsq = cu, total, sumsq


class _LayerFn(torch.autograd.Function):
    """One post-LN transformer block (HF RobertaLayer): fused QKV -> pad-masked attention -> dense ->
    LN(. + x) -> dense+GELU -> dense -> LN(. + x1).  `valid` is the [B, L] pad mask, or a PackedSeqs for pad-free rows."""

    @staticmethod
    def forward(ctx, x, valid, layer, B, L):
        cfg = layer.config
        H, nh = cfg.hidden_size, cfg.num_attention_heads
        ad = x.dtype
        sa = layer.attention.self
        # BASELINE configs[4]: QKV and the two FFN products on the fp8 matrix cores; operands quantised by their producers (ops.Fp8Site)
        fp8 = ops.FP8_FWD[0] and ad == torch.bfloat16 and H >= 256 and H % 64 == 0
        need_bwd = getattr(layer, "_need_bwd", True)
        if fp8:
            qkv = ops.linear_fwd(x, sa.qkv_weight, bias=sa.qkv_bias.data, xq=ops.fp8_take(x))
        else:
            qkv = ops.gemm_nt(x, ops.weight(sa.qkv_weight, ad), bias=sa.qkv_bias.data)
        # dropouts of the HF layer (train mode only): attention probabilities inside the fused kernel, hidden states after the
        # attention-output and the FFN-output dense (before their residual LayerNorms); counter-based masks, replayed in backward
        pa = float(cfg.attention_probs_dropout_prob) if layer.training else 0.0
        ph = float(cfg.hidden_dropout_prob) if layer.training else 0.0
        if pa > 0.0 and not ops._mfma_attn_ok(ops.AttnGeom(1, B, nh, H // nh, L), ad):
            raise RuntimeError("attention-probability dropout needs the matrix-core attention path (bf16); set the rate to 0 for the fp32 mode")
        seeds = (_next_seed(), _next_seed(), _next_seed()) if (pa > 0.0 or ph > 0.0) else (0, 0, 0)
        if isinstance(valid, PackedSeqs):
            geom = ops.AttnGeom(2, B, nh, H // nh, L, 1, valid.total, 0, 0, 1.0 / math.sqrt(H // nh), sumsq=valid.sumsq, drop_p=pa, drop_seed=seeds[0])
            valid = valid.cu
        else:
            geom = ops.AttnGeom(1, B, nh, H // nh, L, 1, 0, 0, 0, 1.0 / math.sqrt(H // nh), drop_p=pa, drop_seed=seeds[0])
        cx, lse = ops.attn_fwd(geom, qkv, valid=valid)
        ao = layer.attention.output
        a = ops.gemm_nt(cx, ops.weight(ao.dense.weight, ad), bias=ao.dense.bias.data)
        if ph > 0.0 and not fp8:          # hidden dropout inside the LayerNorm's pass (same mask / bits as the separate launch)
            x1, mean1, rstd1, s1 = ops.layernorm_dropout_fwd(a, ao.LayerNorm.weight.data, ao.LayerNorm.bias.data, cfg.layer_norm_eps, x, ph, seeds[1])
            x1q = None
        else:
            a = ops.dropout(a, ph, seeds[1])
            x1, mean1, rstd1, s1, x1q = ops.layernorm_fwd(a, ao.LayerNorm.weight.data, ao.LayerNorm.bias.data, cfg.layer_norm_eps, pre=x,
                                                          want_sum=True, emit=ops.fp8_site(layer, "x1", x.device) if fp8 else False)
        it, ot = layer.intermediate, layer.output
        ipre = torch.empty((x.shape[0], cfg.intermediate_size), dtype=ad, device=x.device) if need_bwd else None
        if fp8:
            iact, iq = ops.linear_fwd(x1, it.dense.weight, bias=it.dense.bias.data, epi=ops.gelu_epi(ipre), aux=ipre, xq=x1q,
                                      emit=ops.fp8_site(layer, "h", x.device), need_out=need_bwd)
            o = ops.linear_fwd(iact, ot.dense.weight, bias=ot.dense.bias.data, xq=iq)
        else:
            iact = ops.gemm_nt(x1, ops.weight(it.dense.weight, ad), bias=it.dense.bias.data, epi=ops.gelu_epi(ipre), aux=ipre)
            o = ops.gemm_nt(iact, ops.weight(ot.dense.weight, ad), bias=ot.dense.bias.data)
        if ph > 0.0 and not fp8:
            x2, mean2, rstd2, s2 = ops.layernorm_dropout_fwd(o, ot.LayerNorm.weight.data, ot.LayerNorm.bias.data, cfg.layer_norm_eps, x1, ph, seeds[2])
            x2q = None
        else:
            o = ops.dropout(o, ph, seeds[2])
            x2, mean2, rstd2, s2, x2q = ops.layernorm_fwd(o, ot.LayerNorm.weight.data, ot.LayerNorm.bias.data, cfg.layer_norm_eps, pre=x1,
                                                          want_sum=True,
                                                          emit=ops.fp8_site(layer, "x2", x.device) if (fp8 and getattr(layer, "_q8_next", False)) else False)
        ops.fp8_put(x2, x2q)                # the next layer's QKV product takes it
        ctx.save_for_backward(x, valid, qkv, cx, lse, s1, mean1, rstd1, x1, ipre, iact, s2, mean2, rstd2)
        ctx.layer, ctx.geom, ctx.hdrop = layer, geom, (ph, seeds[1], seeds[2])
        ctx.dgelu_epi = ops.dgelu_epi(ipre)  # what ipre holds: gelu'(pre-activation) (EPI_GELU_DG -> EPI_MUL_AUX) or the pre-activation itself
        return x2

    @staticmethod
    def backward(ctx, g):
        with ops.wgrad_group():               # the layer's four weight gradients: one grouped launch when the group closes
            return _LayerFn._backward(ctx, g)

    @staticmethod
    def _backward(ctx, g):
        x, valid, qkv, cx, lse, s1, mean1, rstd1, x1, ipre, iact, s2, mean2, rstd2 = ctx.saved_tensors
        layer, geom = ctx.layer, ctx.geom
        ad = x.dtype
        sa, ao, it, ot = layer.attention.self, layer.attention.output, layer.intermediate, layer.output
        ph, seed1, seed2 = ctx.hdrop
        # d(dropout(o) + x1) and d(o) = its dropout: one pass (the residual keeps ds2)
        ds2, do = ops.layernorm_bwd_dropout(g.contiguous(), s2, ot.LayerNorm.weight, ot.LayerNorm.bias, mean2, rstd2, ph, seed2)
        ops.linear_wgrad(do, iact, ot.dense.weight, ot.dense.bias)
        dipre = ops.gemm_nt(do, ops.weight_t(ot.dense.weight, ad), epi=ctx.dgelu_epi, aux=ipre)
        ops.linear_wgrad(dipre, x1, it.dense.weight, it.dense.bias)
        g1 = ops.gemm_nt(dipre, ops.weight_t(it.dense.weight, ad), epi=hip.EPI_ADD_AUX, aux=ds2)
        ds1, da = ops.layernorm_bwd_dropout(g1, s1, ao.LayerNorm.weight, ao.LayerNorm.bias, mean1, rstd1, ph, seed1)
        ops.linear_wgrad(da, cx, ao.dense.weight, ao.dense.bias)
        dcx = ops.gemm_nt(da, ops.weight_t(ao.dense.weight, ad))
        dqkv = ops.attn_bwd(geom, qkv, cx, dcx, lse, valid=valid)
        ops.linear_wgrad(dqkv, x, sa.qkv_weight, sa.qkv_bias)
        dx = ops.gemm_nt(dqkv, ops.weight_t(sa.qkv_weight, ad), epi=hip.EPI_ADD_AUX, aux=ds1)
        return dx, None, None, None, None


class _MaskedMeanFn(torch.autograd.Function):
    """(tok * mask).sum(1) / mask.sum(-1)   (unixcoder.py:37)."""

    @staticmethod
    def forward(ctx, tok, valid, B, L):
        H = tok.shape[1]
        out = torch.empty((B, H), dtype=tok.dtype, device=tok.device)
        call("mean_pool_fwd", ptr(tok), ptr(valid), ptr(out), B, L, H, dt(tok))
        ctx.save_for_backward(valid)
        ctx.dims = (B, L, H)
        return out

    @staticmethod
    def backward(ctx, dout):
        (valid,) = ctx.saved_tensors
        B, L, H = ctx.dims
        dx = torch.empty((B * L, H), dtype=dout.dtype, device=dout.device)
        dout = dout.contiguous()
        call("mean_pool_bwd", ptr(dout), ptr(valid), ptr(dx), B, L, H, dt(dx))
        return dx, None, None, None


# ------------------------------------------------------------------------------------------------ modules
class RobertaSelfAttention(nn.Module):
    def __init__(self, config):
        super().__init__()
        H = config.hidden_size
        self.qkv_weight = nn.Parameter(torch.empty(3 * H, H).normal_(0, 0.02))
        self.qkv_bias = nn.Parameter(torch.zeros(3 * H))
        self._register_state_dict_hook(self._split_hook)
        self._register_load_state_dict_pre_hook(self._merge_hook)

    @staticmethod
    def _split_hook(module, state_dict, prefix, local_metadata):
        w = state_dict.pop(prefix + "qkv_weight")
        b = state_dict.pop(prefix + "qkv_bias")
        H = w.shape[1]
        for i, n in enumerate(("query", "key", "value")):
            state_dict[f"{prefix}{n}.weight"] = w[i * H:(i + 1) * H]
            state_dict[f"{prefix}{n}.bias"] = b[i * H:(i + 1) * H]

    def _merge_hook(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        names = ("query", "key", "value")
        if all(f"{prefix}{n}.weight" in state_dict for n in names):
            state_dict[prefix + "qkv_weight"] = torch.cat([state_dict.pop(f"{prefix}{n}.weight") for n in names], 0)
            state_dict[prefix + "qkv_bias"] = torch.cat([state_dict.pop(f"{prefix}{n}.bias") for n in names], 0)


class _DenseLN(nn.Module):
    def __init__(self, fin, fout, eps):
        super().__init__()
        self.dense = nn.Linear(fin, fout)
        self.LayerNorm = nn.LayerNorm(fout, eps=eps)


class _Dense(nn.Module):
    def __init__(self, fin, fout):
        super().__init__()
        self.dense = nn.Linear(fin, fout)


class _Attention(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.self = RobertaSelfAttention(config)
        self.output = _DenseLN(config.hidden_size, config.hidden_size, config.layer_norm_eps)


class RobertaLayer(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.attention = _Attention(config)
        self.intermediate = _Dense(config.hidden_size, config.intermediate_size)
        self.output = _DenseLN(config.intermediate_size, config.hidden_size, config.layer_norm_eps)


class RobertaEmbeddings(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.word_embeddings = nn.Embedding(config.vocab_size, config.hidden_size, padding_idx=config.pad_token_id)
        self.position_embeddings = nn.Embedding(config.max_position_embeddings, config.hidden_size, padding_idx=config.pad_token_id)
        self.token_type_embeddings = nn.Embedding(config.type_vocab_size, config.hidden_size)
        self.LayerNorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        state_dict.pop(prefix + "position_ids", None)          # HF <= 4.30 buffer
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)


class _Encoder(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.layer = nn.ModuleList([RobertaLayer(config) for _ in range(config.num_hidden_layers)])


class RobertaModel(nn.Module):
    """RoBERTa encoder with HF state_dict naming; ``forward(ids) -> (tokens [B,L,H],)`` with the reference's
    pad mask derived from ``ids != pad``."""

    def __init__(self, config, act_dtype=torch.bfloat16):
        super().__init__()
        self.config = config
        self.act_dtype = act_dtype
        self.embeddings = RobertaEmbeddings(config)
        self.encoder = _Encoder(config)
        self.pooler = _Dense(config.hidden_size, config.hidden_size)       # unused by MVulD; kept for checkpoints
        for p in self.pooler.parameters():
            p.requires_grad_(False)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                m.weight.data.normal_(0, 0.02)
                m.bias.data.zero_()
            elif isinstance(m, nn.Embedding):
                m.weight.data.normal_(0, 0.02)

    def encode(self, source_ids):
        """-> (tokens [B*L, H], valid int32 [B, L])"""
        B, L = source_ids.shape
        x, valid = _EmbedFn.apply(self.embeddings.word_embeddings.weight, source_ids, self.embeddings, self.act_dtype)
        x = self._embedding_dropout(x)
        x = self._run_layers(x, valid, B, L)
        return x, valid

    def _embedding_dropout(self, x):
        p = float(self.config.hidden_dropout_prob) if self.training else 0.0      # HF RobertaEmbeddings: LayerNorm, then dropout
        return _DropoutFn.apply(x, p, _next_seed()) if p > 0.0 else x

    def can_pack(self):
        return self.act_dtype == torch.bfloat16 and ops.ATTN_IMPL[0] == "auto"

    @staticmethod
    def pack_plan(seq_lens, device, L):
        """PackedSeqs (device cu_seqlens + host totals) from host-side per-sequence token counts.  Build it once outside a
        hipGraph capture and pass it as `seq_lens`: the forward then does no host-side work."""
        lens = torch.as_tensor(seq_lens, dtype=torch.int64).view(-1).cpu()
        assert int(lens.max()) <= L and int(lens.min()) >= 0, "seq_lens out of range"
        cu_host = torch.zeros(lens.numel() + 1, dtype=torch.int32)
        cu_host[1:] = torch.cumsum(lens, 0)
        T = int(cu_host[-1])
        assert T > 0, "every sequence is empty"
        return PackedSeqs(_CU_STAGE.stage(cu_host, torch.device(device)), T, float((lens * lens).sum()))

    def encode_packed(self, source_ids, seq_lens):
        """Pad-free encoder pass.  seq_lens: per-sequence count of non-pad tokens, a HOST int tensor / list (the data loader
        has the ids on the host anyway: counting there avoids a device -> host sync here).
        -> (tokens [T, H] packed, PackedSeqs, rowmap int32 [T] = padded row b*L + l of every packed row)"""
        B, L = source_ids.shape
        packed = seq_lens if isinstance(seq_lens, PackedSeqs) else self.pack_plan(seq_lens, source_ids.device, L)
        assert packed.cu.numel() == B + 1, "seq_lens does not match source_ids"
        T = packed.total
        x, rowmap = _PackEmbedFn.apply(self.embeddings.word_embeddings.weight, source_ids, packed.cu, T, self.embeddings, self.act_dtype)
        x = self._embedding_dropout(x)
        x = self._run_layers(x, packed, B, L)
        return x, packed, rowmap

    def _run_layers(self, x, valid, B, L):
        if ops.FP8_FWD[0] and not ops.FP8_IN_FUSED[0]:
            ops.fp8_roll(x.device)
        n = len(self.encoder.layer)
        for i, layer in enumerate(self.encoder.layer):
            layer._need_bwd, layer._q8_next = torch.is_grad_enabled(), i + 1 < n
            x = _LayerFn.apply(x, valid, layer, B, L)
        return x

    def forward(self, source_ids, attention_mask=None):
        B, L = source_ids.shape
        x, _ = self.encode(source_ids)
        return (x.view(B, L, -1),)


class MyUniXcoder(nn.Module):
    def __init__(self, encoder, config, tokenizer=None, tokenize=None):
        super().__init__()
        self.encoder = encoder
        self.config = config
        self.tokenizer = tokenizer
        self.tokenize = tokenize
        self.classifier = nn.Linear(config.hidden_size, 2)
        self.max_source_length = 512
        self.return_tokens = True        # False: get_xcode_vec skips unpacking the token embeddings (the fused model reads only `sent`)

    def get_xcode_vec(self, source_ids, seq_lens=None):
        """Token embeddings [B,L,H] and sentence embeddings [B,H] (masked mean over non-pad tokens).
        With `seq_lens` (host-side counts of non-pad tokens per sequence) the encoder runs pad-free on the packed tokens:
        identical sentence vectors and non-pad token rows; the pad rows of the token output are zero."""
        B, L = source_ids.shape
        if seq_lens is not None and self.encoder.can_pack():
            tok, packed, rowmap = self.encoder.encode_packed(source_ids, seq_lens)
            sent = _SegmentMeanFn.apply(tok, packed.cu, B)
            full = _UnpackFn.apply(tok, rowmap, B * L).view(B, L, -1) if self.return_tokens else None
            return full, sent
        tok, valid = self.encoder.encode(source_ids)
        sent = _MaskedMeanFn.apply(tok, valid, B, L)
        return tok.view(B, L, -1), sent

    def forward(self, source_ids=None, labels=None):
        from .GraphModel import linear_act, cross_entropy
        source_ids = source_ids.view(-1, self.max_source_length)
        _, vec = self.get_xcode_vec(source_ids)
        logits = linear_act(vec, self.classifier.weight, self.classifier.bias, act=None, out_dtype=torch.float32)
        if labels is not None:
            loss, prob = cross_entropy(logits, labels)
            return loss, prob
        _, prob = cross_entropy(logits, torch.zeros(logits.shape[0], dtype=torch.int64, device=logits.device))
        return prob

    def get_repr(self, input_ids, labels=None):
        source_ids = input_ids.view(-1, self.max_source_length)
        _, vec = self.get_xcode_vec(source_ids)
        return vec, labels

    def encode_lines(self, line_ids, line_lens=None, chunk=8192):
        """Node embeddings of a graph: one sentence vector per source line (the rows `myEncode` produces for data_list.py:293-299,
        cached offline by the reference), from the lines' token ids [n_lines, L].  With `line_lens` (host-side non-pad counts) the
        lines are packed: a batch of 200 lines of ~15 tokens costs what 6 padded 512-token rows would.  No gradients (the reference
        computes them offline under torch.no_grad)."""
        outs = []
        with torch.no_grad():
            for a in range(0, line_ids.shape[0], chunk):
                ids = line_ids[a:a + chunk]
                lens = None if line_lens is None else torch.as_tensor(line_lens)[a:a + chunk]
                keep, self.return_tokens = self.return_tokens, False
                try:
                    _, vec = self.get_xcode_vec(ids, lens)
                finally:
                    self.return_tokens = keep
                outs.append(vec)
        return torch.cat(outs, 0) if len(outs) > 1 else outs[0]

    def myEncode(self, sents: list):
        if self.tokenize is None:
            raise RuntimeError("myEncode needs the UniXcoder tokenizer (not available offline); pass token ids to get_repr")
        rows = [self.tokenize([' '.join(s.split())], max_length=512, padding=True)[0] for s in sents]
        ids = torch.tensor(rows, dtype=torch.long, device=next(self.parameters()).device)
        vec, _ = self.get_repr(ids)
        return vec


class UniXcoder(nn.Module):
    """``UniXcoder(model_name)``: builds the encoder from a local config.  ``model_name`` may be a
    RobertaConfigLite, a dict of its fields, or a hub name (then the unixcoder-base-nine shape is assumed: there
    is no network, so weights are random-init and no tokenizer is attached)."""

    def __init__(self, model_name="microsoft/unixcoder-base-nine", act_dtype=torch.bfloat16):
        super().__init__()
        if isinstance(model_name, RobertaConfigLite):
            self.config = model_name
        elif isinstance(model_name, dict):
            self.config = RobertaConfigLite(**model_name)
        else:
            self.config = RobertaConfigLite()
        self.model = RobertaModel(self.config, act_dtype)
        self.tokenizer = None

    def tokenize(self, inputs, mode="<encoder-only>", max_length=512, padding=False):
        raise RuntimeError("UniXcoder tokenizer files are not available offline; feed token ids")

    def forward(self, source_ids):
        B, L = source_ids.shape
        tok, valid = self.model.encode(source_ids)
        return tok.view(B, L, -1), _MaskedMeanFn.apply(tok, valid, B, L)
