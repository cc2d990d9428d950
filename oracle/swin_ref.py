"""Oracle: SwinV2 forward_features, functional over a state_dict.  TEST INFRASTRUCTURE.

Follows /root/reference/mvuld/models/swin_transformer_v2.py:
  PatchEmbed.forward            :485-493
  SwinTransformerBlock.forward  :270-306  (mask construction :245-264)
  WindowAttention.forward       :140-179  (coords table :97-113, rel index :115-126)
  Mlp.forward                   :26-32
  PatchMerging.forward          :343-364
  SwinTransformerV2.forward_features :623-635
Dropout / DropPath are identity (eval or rate 0).  Buffers are rebuilt from the
geometry rather than read from the state_dict.
"""
import math
from dataclasses import dataclass, field
from typing import List

import torch
import torch.nn.functional as F


@dataclass
class SwinCfg:
    img_size: int = 448
    patch_size: int = 4
    in_chans: int = 3
    embed_dim: int = 128
    depths: List[int] = field(default_factory=lambda: [2, 2, 18, 2])
    num_heads: List[int] = field(default_factory=lambda: [4, 8, 16, 32])
    window_size: int = 28
    mlp_ratio: float = 4.0
    pretrained_window_sizes: List[int] = field(default_factory=lambda: [12, 12, 12, 6])
    num_classes: int = 2

    def stage_geometry(self, i):
        """(resolution, dim, heads, window, [shift per block]) of stage i, after the
        clamp at swin_transformer_v2.py:228-232."""
        res = self.img_size // self.patch_size // (2 ** i)
        dim = self.embed_dim * (2 ** i)
        ws = self.window_size
        shifts = []
        for j in range(self.depths[i]):
            s = 0 if j % 2 == 0 else self.window_size // 2
            if res <= self.window_size:
                s = 0
            shifts.append(s)
        if res <= self.window_size:
            ws = res
        return res, dim, self.num_heads[i], ws, shifts


def coords_table(ws: int, pretrained_ws: int) -> torch.Tensor:
    """[(2ws-1)^2, 2] log-spaced relative coordinates (:97-113)."""
    r = torch.arange(-(ws - 1), ws, dtype=torch.float32)
    gy, gx = torch.meshgrid(r, r, indexing="ij")
    t = torch.stack([gy, gx], dim=-1)
    denom = (pretrained_ws - 1) if pretrained_ws > 0 else (ws - 1)
    t = t / denom * 8
    t = torch.sign(t) * torch.log2(torch.abs(t) + 1.0) / math.log2(8)
    return t.reshape(-1, 2)


def rel_index(ws: int) -> torch.Tensor:
    """[ws*ws, ws*ws] index into the (2ws-1)^2 table (:115-126)."""
    ys, xs = torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij")
    ys, xs = ys.reshape(-1), xs.reshape(-1)
    dy = ys[:, None] - ys[None, :] + ws - 1
    dx = xs[:, None] - xs[None, :] + ws - 1
    return dy * (2 * ws - 1) + dx


def shift_mask(res: int, ws: int, shift: int) -> torch.Tensor:
    """[nW, ws*ws, ws*ws] 0/-100 mask of the shifted configuration (:245-264)."""
    ids = torch.zeros(res, res)
    bounds = [(0, res - ws), (res - ws, res - shift), (res - shift, res)]
    c = 0
    for (h0, h1) in bounds:
        for (w0, w1) in bounds:
            ids[h0:h1, w0:w1] = c
            c += 1
    n = res // ws
    m = ids.view(n, ws, n, ws).permute(0, 2, 1, 3).reshape(n * n, ws * ws)
    d = m[:, None, :] - m[:, :, None]
    return torch.where(d != 0, torch.full_like(d, -100.0), torch.zeros_like(d))


def window_attention(sd, p, xw, heads, ws, pws, mask):
    """xw: [B_, N, C] windows (:140-179)."""
    B_, N, C = xw.shape
    bias = torch.cat([sd[p + "q_bias"], torch.zeros_like(sd[p + "v_bias"]), sd[p + "v_bias"]])
    qkv = F.linear(xw, sd[p + "qkv.weight"], bias).reshape(B_, N, 3, heads, C // heads).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    attn = F.normalize(q, dim=-1) @ F.normalize(k, dim=-1).transpose(-2, -1)
    scale = torch.clamp(sd[p + "logit_scale"], max=math.log(1.0 / 0.01)).exp()
    attn = attn * scale
    tab = coords_table(ws, pws).to(xw.dtype)
    hid = F.relu(F.linear(tab, sd[p + "cpb_mlp.0.weight"], sd[p + "cpb_mlp.0.bias"]))
    tab = F.linear(hid, sd[p + "cpb_mlp.2.weight"])                     # [(2ws-1)^2, H]
    rpb = tab[rel_index(ws).reshape(-1)].view(N, N, heads).permute(2, 0, 1)
    attn = attn + 16 * torch.sigmoid(rpb).unsqueeze(0)
    if mask is not None:
        nW = mask.shape[0]
        attn = attn.view(B_ // nW, nW, heads, N, N) + mask[None, :, None]
        attn = attn.view(B_, heads, N, N)
    attn = attn.softmax(dim=-1)
    out = (attn @ v).transpose(1, 2).reshape(B_, N, C)
    return F.linear(out, sd[p + "proj.weight"], sd[p + "proj.bias"])


def swin_block(sd, p, x, res, dim, heads, ws, shift, pws):
    B, L, C = x.shape
    shortcut = x
    xs = x.view(B, res, res, C)
    if shift > 0:
        xs = torch.roll(xs, shifts=(-shift, -shift), dims=(1, 2))
    n = res // ws
    xw = xs.view(B, n, ws, n, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(B * n * n, ws * ws, C)
    mask = shift_mask(res, ws, shift).to(x.dtype) if shift > 0 else None
    aw = window_attention(sd, p + "attn.", xw, heads, ws, pws, mask)
    xs = aw.view(B, n, n, ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(B, res, res, C)
    if shift > 0:
        xs = torch.roll(xs, shifts=(shift, shift), dims=(1, 2))
    x = xs.reshape(B, L, C)
    x = shortcut + F.layer_norm(x, (C,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-5)
    h = F.linear(x, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"])
    h = F.gelu(h)
    h = F.linear(h, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
    return x + F.layer_norm(h, (C,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-5)


def patch_merging(sd, p, x, res):
    B, L, C = x.shape
    x = x.view(B, res, res, C)
    x = torch.cat([x[:, 0::2, 0::2], x[:, 1::2, 0::2], x[:, 0::2, 1::2], x[:, 1::2, 1::2]], -1)
    x = x.view(B, -1, 4 * C)
    x = F.linear(x, sd[p + "reduction.weight"])
    return F.layer_norm(x, (2 * C,), sd[p + "norm.weight"], sd[p + "norm.bias"], 1e-5)


def swin_tokens(sd, x, cfg: SwinCfg, prefix="", upto_stage=None):
    """Token tensor after the last stage (before final norm)."""
    w = sd[prefix + "patch_embed.proj.weight"]
    x = F.conv2d(x, w, sd[prefix + "patch_embed.proj.bias"], stride=cfg.patch_size)
    x = x.flatten(2).transpose(1, 2)
    x = F.layer_norm(x, (cfg.embed_dim,), sd[prefix + "patch_embed.norm.weight"],
                     sd[prefix + "patch_embed.norm.bias"], 1e-5)
    nst = len(cfg.depths) if upto_stage is None else upto_stage
    for i in range(nst):
        res, dim, heads, ws, shifts = cfg.stage_geometry(i)
        for j, s in enumerate(shifts):
            x = swin_block(sd, f"{prefix}layers.{i}.blocks.{j}.", x, res, dim, heads, ws, s,
                           cfg.pretrained_window_sizes[i])
        if i < len(cfg.depths) - 1:
            x = patch_merging(sd, f"{prefix}layers.{i}.downsample.", x, res)
    return x


def swin_forward_features(sd, x, cfg: SwinCfg, prefix=""):
    """[B,3,S,S] -> [B, embed_dim*8]   (:623-635)."""
    x = swin_tokens(sd, x, cfg, prefix)
    C = x.shape[-1]
    x = F.layer_norm(x, (C,), sd[prefix + "norm.weight"], sd[prefix + "norm.bias"], 1e-5)
    return x.mean(dim=1)


def swin_param_shapes(cfg: SwinCfg, prefix=""):
    """name -> shape of every parameter, in the reference's state_dict naming."""
    P = {}
    E = cfg.embed_dim
    P["patch_embed.proj.weight"] = (E, cfg.in_chans, cfg.patch_size, cfg.patch_size)
    P["patch_embed.proj.bias"] = (E,)
    P["patch_embed.norm.weight"] = (E,)
    P["patch_embed.norm.bias"] = (E,)
    for i, d in enumerate(cfg.depths):
        C = E * 2 ** i
        H = cfg.num_heads[i]
        hid = int(C * cfg.mlp_ratio)
        for j in range(d):
            b = f"layers.{i}.blocks.{j}."
            P[b + "norm1.weight"] = (C,); P[b + "norm1.bias"] = (C,)
            P[b + "attn.logit_scale"] = (H, 1, 1)
            P[b + "attn.cpb_mlp.0.weight"] = (512, 2); P[b + "attn.cpb_mlp.0.bias"] = (512,)
            P[b + "attn.cpb_mlp.2.weight"] = (H, 512)
            P[b + "attn.qkv.weight"] = (3 * C, C)
            P[b + "attn.q_bias"] = (C,); P[b + "attn.v_bias"] = (C,)
            P[b + "attn.proj.weight"] = (C, C); P[b + "attn.proj.bias"] = (C,)
            P[b + "norm2.weight"] = (C,); P[b + "norm2.bias"] = (C,)
            P[b + "mlp.fc1.weight"] = (hid, C); P[b + "mlp.fc1.bias"] = (hid,)
            P[b + "mlp.fc2.weight"] = (C, hid); P[b + "mlp.fc2.bias"] = (C,)
        if i < len(cfg.depths) - 1:
            b = f"layers.{i}.downsample."
            P[b + "reduction.weight"] = (2 * C, 4 * C)
            P[b + "norm.weight"] = (2 * C,); P[b + "norm.bias"] = (2 * C,)
    CF = E * 2 ** (len(cfg.depths) - 1)
    P["norm.weight"] = (CF,); P["norm.bias"] = (CF,)
    P["head.weight"] = (cfg.num_classes, CF); P["head.bias"] = (cfg.num_classes,)
    return {prefix + k: v for k, v in P.items()}
