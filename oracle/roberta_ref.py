"""Oracle: UniXcoder encoder (RoBERTa-base shape) + masked mean pool.  TEST INFRASTRUCTURE.

The arithmetic lives in HF transformers==4.18.0 ``RobertaModel`` (third party,
absent from /root/reference; environment.yml:289).  Restated here from its
published algorithm, anchored on the reference call site
/root/reference/mvuld/models/unixcoder.py:33-38 (``get_xcode_vec``):

    mask = ids != pad
    tok  = encoder(ids, attention_mask = mask[:,None,:] * mask[:,:,None])[0]
    sent = (tok * mask[...,None]).sum(1) / mask.sum(-1)[...,None]

4.18.0 semantics of a 3-D mask: broadcast to [B,1,L,L] and turned into the
additive term (1 - m) * -10000 (no causal mask even with is_decoder=True, which
unixcoder.py:109 sets).  Position ids: cumsum(ids != pad) * (ids != pad) + pad.
Post-LN blocks, GELU(erf), LayerNorm eps from config (1e-5), scale 1/sqrt(64).
Dropout is identity (eval or p = 0).  **Parity unpinned** by any reference test;
tests/test_cpu_oracle_and_host.py::test_oracle_roberta_matches_golden checks this file against fixtures generated
(tests/golden/make_golden.py) from the installed transformers RobertaModel driven with the equivalent 4-D additive mask.
"""
import math
from dataclasses import dataclass

import torch
import torch.nn.functional as F


@dataclass
class RobertaCfg:
    vocab_size: int = 51416
    hidden_size: int = 768
    num_layers: int = 12
    num_heads: int = 12
    intermediate_size: int = 3072
    max_position: int = 1026
    type_vocab_size: int = 10
    pad_token_id: int = 1
    ln_eps: float = 1e-5


def position_ids(ids, pad):
    m = ids.ne(pad).long()
    return torch.cumsum(m, dim=1) * m + pad


def roberta_encode(sd, ids, cfg: RobertaCfg, prefix="encoder."):
    """ids [B,L] int64 -> token embeddings [B,L,H] with the reference's 3-D pad mask."""
    B, L = ids.shape
    H, nh = cfg.hidden_size, cfg.num_heads
    hd = H // nh
    mask = ids.ne(cfg.pad_token_id)
    e = prefix + "embeddings."
    x = sd[e + "word_embeddings.weight"][ids] \
        + sd[e + "position_embeddings.weight"][position_ids(ids, cfg.pad_token_id)] \
        + sd[e + "token_type_embeddings.weight"][torch.zeros_like(ids)]
    x = F.layer_norm(x, (H,), sd[e + "LayerNorm.weight"], sd[e + "LayerNorm.bias"], cfg.ln_eps)
    m3 = (mask[:, None, :] & mask[:, :, None]).to(x.dtype)             # [B,L,L]
    add = (1.0 - m3[:, None]) * -10000.0                               # [B,1,L,L]
    for i in range(cfg.num_layers):
        p = f"{prefix}encoder.layer.{i}."
        a = p + "attention.self."
        q = F.linear(x, sd[a + "query.weight"], sd[a + "query.bias"]).view(B, L, nh, hd).transpose(1, 2)
        k = F.linear(x, sd[a + "key.weight"], sd[a + "key.bias"]).view(B, L, nh, hd).transpose(1, 2)
        v = F.linear(x, sd[a + "value.weight"], sd[a + "value.bias"]).view(B, L, nh, hd).transpose(1, 2)
        s = q @ k.transpose(-1, -2) / math.sqrt(hd) + add
        ctx = (s.softmax(-1) @ v).transpose(1, 2).reshape(B, L, H)
        o = p + "attention.output."
        y = F.linear(ctx, sd[o + "dense.weight"], sd[o + "dense.bias"])
        x = F.layer_norm(y + x, (H,), sd[o + "LayerNorm.weight"], sd[o + "LayerNorm.bias"], cfg.ln_eps)
        h = F.gelu(F.linear(x, sd[p + "intermediate.dense.weight"], sd[p + "intermediate.dense.bias"]))
        y = F.linear(h, sd[p + "output.dense.weight"], sd[p + "output.dense.bias"])
        x = F.layer_norm(y + x, (H,), sd[p + "output.LayerNorm.weight"], sd[p + "output.LayerNorm.bias"],
                         cfg.ln_eps)
    return x, mask


def unixcoder_sentence(sd, ids, cfg: RobertaCfg, prefix="encoder."):
    """(token [B,L,H], sentence [B,H]) as MyUniXcoder.get_xcode_vec (unixcoder.py:33-38)."""
    tok, mask = roberta_encode(sd, ids, cfg, prefix)
    m = mask.to(tok.dtype)
    sent = (tok * m[..., None]).sum(1) / m.sum(-1)[..., None]
    return tok, sent


def roberta_param_shapes(cfg: RobertaCfg, prefix="encoder.", with_pooler=True):
    H, I = cfg.hidden_size, cfg.intermediate_size
    P = {}
    e = "embeddings."
    P[e + "word_embeddings.weight"] = (cfg.vocab_size, H)
    P[e + "position_embeddings.weight"] = (cfg.max_position, H)
    P[e + "token_type_embeddings.weight"] = (cfg.type_vocab_size, H)
    P[e + "LayerNorm.weight"] = (H,); P[e + "LayerNorm.bias"] = (H,)
    for i in range(cfg.num_layers):
        p = f"encoder.layer.{i}."
        for n in ("query", "key", "value"):
            P[p + f"attention.self.{n}.weight"] = (H, H); P[p + f"attention.self.{n}.bias"] = (H,)
        P[p + "attention.output.dense.weight"] = (H, H); P[p + "attention.output.dense.bias"] = (H,)
        P[p + "attention.output.LayerNorm.weight"] = (H,); P[p + "attention.output.LayerNorm.bias"] = (H,)
        P[p + "intermediate.dense.weight"] = (I, H); P[p + "intermediate.dense.bias"] = (I,)
        P[p + "output.dense.weight"] = (H, I); P[p + "output.dense.bias"] = (H,)
        P[p + "output.LayerNorm.weight"] = (H,); P[p + "output.LayerNorm.bias"] = (H,)
    if with_pooler:
        P["pooler.dense.weight"] = (H, H); P["pooler.dense.bias"] = (H,)
    return {prefix + k: v for k, v in P.items()}
