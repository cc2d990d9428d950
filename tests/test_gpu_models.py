"""GPU parity of the three modules and the fused model against the CPU oracle / committed goldens.

Tolerances follow BASELINE.json's north_star: logits within 1e-3 (fp32) / 1e-2 (bf16) of the reference."""
import types

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from util import rel, rel_l2, golden, load_synth_into, synth

pytestmark = pytest.mark.gpu
DTYPES = [torch.float32, torch.bfloat16]

SWIN_MINI = dict(img_size=224, embed_dim=32, depths=[2, 2, 2, 2], num_heads=[1, 2, 4, 8], window_size=14,
                 pretrained_window_sizes=[12, 12, 12, 6])
SWIN_SMALL = dict(img_size=448, embed_dim=32, depths=[2, 2, 2, 2], num_heads=[1, 2, 4, 8], window_size=28,
                  pretrained_window_sizes=[12, 12, 12, 6])
SWIN_BASE = dict(img_size=448, embed_dim=128, depths=[2, 2, 18, 2], num_heads=[4, 8, 16, 32], window_size=28,
                 pretrained_window_sizes=[12, 12, 12, 6])


def _swin(kw, dtype, gpu):
    from mvuld_amd.models.swin_transformer_v2 import SwinTransformerV2
    m = SwinTransformerV2(num_classes=2, drop_path_rate=0.2, act_dtype=dtype, **kw)
    sd, _ = load_synth_into(m)
    return m.to(gpu), sd


def _images(n, size, first=1000):
    from mvuld_amd.data import synthetic
    return torch.stack([synthetic.make_image(first + i, size) for i in range(n)])


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("name,kw,B", [("swin_mini224", SWIN_MINI, 2), ("swin_small448", SWIN_SMALL, 1)])
def test_swin_features_vs_golden(gpu, dtype, name, kw, B):
    m, _ = _swin(kw, dtype, gpu)
    m.eval()
    x = _images(B, kw["img_size"])
    with torch.no_grad():
        f = m.forward_features(x.to(gpu))
    ref = torch.from_numpy(golden(name)["feat"])
    e = rel(f, ref)
    print(f"[{name} {dtype}] rel err vs reference golden = {e:.3e}")
    assert e < (1e-3 if dtype == torch.float32 else 3e-2)


@pytest.mark.parametrize("dtype", DTYPES)
def test_swin_base_vs_golden(gpu, dtype):
    m, _ = _swin(SWIN_BASE, dtype, gpu)
    m.eval()
    with torch.no_grad():
        f = m.forward_features(_images(1, 448).to(gpu))
    ref = torch.from_numpy(golden("swin_base448")["feat"])
    e = rel(f, ref)
    print(f"[swin_base448 {dtype}] rel err vs reference golden = {e:.3e}")
    assert e < (1e-3 if dtype == torch.float32 else 3e-2)


@pytest.mark.parametrize("dtype", DTYPES)
def test_swin_gradients_vs_oracle(gpu, dtype):
    """train mode, drop-path 0: d(sum(feat*w))/d(params) against autograd through the oracle."""
    from oracle import swin_ref
    kw = dict(SWIN_MINI)
    from mvuld_amd.models.swin_transformer_v2 import SwinTransformerV2
    m = SwinTransformerV2(num_classes=2, drop_path_rate=0.0, act_dtype=dtype, **kw)
    sd, _ = load_synth_into(m)
    m = m.to(gpu).train()
    x = _images(2, 224)
    wv = synth.tensor("swin/gradw", (2, 256))
    cfg = swin_ref.SwinCfg(img_size=224, embed_dim=32, depths=[2, 2, 2, 2], num_heads=[1, 2, 4, 8], window_size=14)
    sdr = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    fr = swin_ref.swin_forward_features(sdr, x, cfg)
    (fr * wv).sum().backward()
    f = m.forward_features(x.to(gpu))
    (f.float() * wv.to(gpu)).sum().backward()
    assert rel(f, fr) < (1e-3 if dtype == torch.float32 else 3e-2)
    worst = []
    for n, p in m.named_parameters():
        if n.startswith("head."):
            continue
        g_ref = sdr[n].grad
        assert p.grad is not None, n
        worst.append((rel_l2(p.grad, g_ref), n))
    worst.sort(reverse=True)
    print(f"[swin grads {dtype}] worst: " + ", ".join(f"{n}={e:.2e}" for e, n in worst[:6]))
    # bf16: the cpb-MLP / logit_scale gradients are sums of O(1e5) bf16-noisy terms that largely cancel
    lim = 2e-3 if dtype == torch.float32 else 1.5e-1
    assert worst[0][0] < lim, worst[:6]


ROB_TINY = dict(vocab_size=1000, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=512,
                max_position_embeddings=130)


def _rob_ids(vocab, L, lens, tag, pad=1):
    rows = []
    for i, n in enumerate(lens):
        r = synth.ints(f"{tag}/{i}", (L,), 5, vocab)
        r[0], r[1], r[2] = 0, 6, 2
        r[n - 1] = 2
        r[n:] = pad
        rows.append(r)
    return torch.stack(rows)


def _unix(kw, dtype, gpu):
    from mvuld_amd.models.unixcoder import RobertaConfigLite, RobertaModel, MyUniXcoder
    rc = RobertaConfigLite(**kw)
    m = MyUniXcoder(RobertaModel(rc, dtype), rc)
    sd, _ = load_synth_into(m)
    return m.to(gpu), sd, rc


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("name,kw,L,lens", [("roberta_tiny", ROB_TINY, 128, [128, 77, 5]), ("roberta_base512", {}, 512, [512, 301])])
def test_unixcoder_sentence_vs_golden(gpu, dtype, name, kw, L, lens):
    m, _, rc = _unix(kw, dtype, gpu)
    m.eval()
    ids = _rob_ids(rc.vocab_size, L, lens, name)
    with torch.no_grad():
        _, sent = m.get_xcode_vec(ids.to(gpu))
    ref = torch.from_numpy(golden(name)["sent"])
    e = rel(sent, ref)
    print(f"[{name} {dtype}] rel err vs golden = {e:.3e}")
    assert e < (1e-3 if dtype == torch.float32 else 3e-2)


@pytest.mark.parametrize("dtype", DTYPES)
def test_unixcoder_gradients_vs_oracle(gpu, dtype):
    from oracle import roberta_ref
    m, sd, rc = _unix(ROB_TINY, dtype, gpu)
    m.train()
    ids = _rob_ids(rc.vocab_size, 128, [128, 77, 5], "roberta_tiny")
    wv = synth.tensor("rob/gradw", (3, 128))
    cfg = roberta_ref.RobertaCfg(vocab_size=1000, hidden_size=128, num_layers=2, num_heads=2, intermediate_size=512, max_position=130)
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    _, sr = roberta_ref.unixcoder_sentence(sdr, ids, cfg)
    (sr * wv).sum().backward()
    _, s = m.get_xcode_vec(ids.to(gpu))
    (s.float() * wv.to(gpu)).sum().backward()
    assert rel(s, sr) < (1e-3 if dtype == torch.float32 else 3e-2)
    state = m.state_dict(keep_vars=False)
    worst = []
    for n, p in m.named_parameters():
        if not p.requires_grad or n.startswith("classifier"):
            continue
        if n.endswith("qkv_weight") or n.endswith("qkv_bias"):
            base = n[:-len("qkv_weight")] if n.endswith("qkv_weight") else n[:-len("qkv_bias")]
            suf = "weight" if n.endswith("qkv_weight") else "bias"
            g_ref = torch.cat([sdr[f"{base}{q}.{suf}"].grad for q in ("query", "key", "value")], 0)
        else:
            g_ref = sdr[n].grad
        worst.append((rel_l2(p.grad, g_ref), n))
    worst.sort(reverse=True)
    print(f"[unixcoder grads {dtype}] worst: " + ", ".join(f"{n}={e:.2e}" for e, n in worst[:6]))
    assert worst[0][0] < (2e-3 if dtype == torch.float32 else 8e-2), worst[:6]


HEAD_NODES = [60, 100, 130, 217]


def _head_inputs():
    from mvuld_amd.data import synthetic
    from mvuld_amd.graph import batch
    g = batch([synthetic.make_graph(2000 + i, n, n) for i, n in enumerate(HEAD_NODES)])
    img = synth.tensor("head/img", (len(HEAD_NODES), 1024), -1, 1)
    txt = synth.tensor("head/txt", (len(HEAD_NODES), 768), -1, 1)
    return g, img, txt


def _head(dtype, gpu):
    from mvuld_amd.models.GraphModel import Multi_DefectModel_new_GCN
    cfg = types.SimpleNamespace(MODEL=types.SimpleNamespace(NUM_CLASSES=2))
    m = Multi_DefectModel_new_GCN(cfg, act_dtype=dtype)
    m.p_gat = m.p_mlp = m.p_hidden = 0.0
    m.gat.feat_drop_p = m.gat2.feat_drop_p = 0.0
    sd, _ = load_synth_into(m)
    return m.to(gpu), sd


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("mode", ["eval", "train"])
def test_head_logits_vs_golden(gpu, dtype, mode):
    m, _ = _head(dtype, gpu)
    m.train(mode == "train")
    g, img, txt = _head_inputs()
    with torch.no_grad():
        lg = m(g.to(gpu), img.to(gpu), txt.to(gpu))
    ref = torch.from_numpy(golden("head")[f"logits_{mode}"])
    err = float((lg.float().cpu() - ref).abs().max())
    print(f"[head {mode} {dtype}] max abs logit err vs reference golden = {err:.3e}")
    # train mode normalises with the statistics of a batch of FOUR: bf16 rounding of the inputs is amplified by
    # 1/std over 4 samples, so the bf16 bound is looser there; eval mode carries the north_star 1e-2 bound.
    lim = 1e-3 if dtype == torch.float32 else (1e-2 if mode == "eval" else 1e-1)
    assert err < lim


@pytest.mark.parametrize("dtype", DTYPES)
def test_head_gradients_vs_oracle(gpu, dtype):
    from oracle import head_ref
    m, sd = _head(dtype, gpu)
    m.train()
    g, img, txt = _head_inputs()
    tgt = torch.tensor([0, 1, 1, 0])
    sdr = {k: v.clone().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in sd.items()}
    ir, tr = img.clone().requires_grad_(True), txt.clone().requires_grad_(True)
    lr_ = head_ref.head_forward(sdr, g.src, g.dst, g.batch_num_nodes(), g.ndata["_UNIX_NODE_EMB"], g.ndata["pos_emb"], ir, tr, training=True)
    F.cross_entropy(lr_, tgt).backward()
    from mvuld_amd.models.GraphModel import cross_entropy
    ig, tg = img.to(gpu).requires_grad_(True), txt.to(gpu).requires_grad_(True)
    lg = m(g.to(gpu), ig, tg)
    loss, _ = cross_entropy(lg, tgt.to(gpu))
    loss.backward()
    assert float((lg.float().cpu() - lr_.detach()).abs().max()) < (1e-3 if dtype == torch.float32 else 2e-2)
    worst = [(rel_l2(ig.grad, ir.grad), "img"), (rel_l2(tg.grad, tr.grad), "txt")]
    for n, p in m.named_parameters():
        if n.startswith(m.unused_parameter_prefixes):
            continue
        assert p.grad is not None, f"no HIP grad for {n}"
        assert sdr[n].grad is not None, f"no oracle grad for {n}"
        worst.append((rel_l2(p.grad, sdr[n].grad), n))
    worst.sort(reverse=True)
    print(f"[head grads {dtype}] worst: " + ", ".join(f"{n}={e:.2e}" for e, n in worst[:8]))
    assert worst[0][0] < (5e-3 if dtype == torch.float32 else 1.5e-1), worst[:8]


def test_rs_gcn_reference_layout(gpu):
    from mvuld_amd.models.Rs_GCN import Rs_GCN
    m = Rs_GCN(512, 512)
    load_synth_into(m, "Rs_GCN_1.")
    m = m.to(gpu)
    v = synth.tensor("rsgcn/in", (4, 512, 100), -1, 1)
    gd = golden("rs_gcn")
    for mode in ("eval", "train"):
        load_synth_into(m, "Rs_GCN_1.")
        m.train(mode == "train")
        with torch.no_grad():
            y, R = m(v.to(gpu))
        assert rel(y, torch.from_numpy(gd[f"y_{mode}"])) < 1e-4
        assert rel(R, torch.from_numpy(gd[f"R_{mode}"])) < 1e-4
