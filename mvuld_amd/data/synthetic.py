"""Synthetic Big-Vul-shaped samples: there is no dataset, Joern or
image renderer on the box, so every benchmark / test input is generated here,
deterministically from the sample index (rank-independent).

Per function:
  image   [3,S,S] ~ U(-1.7,1.7) (unit variance, post-Normalize statistics)
  ids     [512] int64: ``<s> <encoder-only> </s>`` (0, 6, 2) then tokens uniform
          in [5, vocab), ``</s>``, then pad id 1; valid length ~ U[128, 512]
  graph   N ~ U{lo..hi} line-level nodes; directed edges = (N-1) random-tree "AST"
          + N "CFG" chain/branch + N random "CDG" (duplicates allowed) + N self-loops
          => E = 4N-1; ``_UNIX_NODE_EMB`` [N,768] ~ 0.5*U(-1.7,1.7), ``pos_emb`` [N,4]
          ~ U(0,1) with ~10 % rows zeroed (OCR misses), ``_FUNC_EMB`` filled by the caller
  label   Bernoulli(0.5)

Mirrors the schema ImageList.item / __getitem__ produce in the reference
(mvuld/data/data_list.py:107-141,265-317).
"""
import numpy as np
import torch

from .. import synth
from ..graph import BatchedGraph, batch as batch_graphs


def make_graph(index: int, n_lo=150, n_hi=250, emb=768, salt=0) -> BatchedGraph:
    tag = f"graph/{index}"
    n = int(synth.ints(tag + "/n", (1,), n_lo, n_hi + 1, salt)[0])
    # AST: node i>0 hangs under a random earlier node
    par = (synth.unit(tag + "/ast", n - 1, salt).astype(np.float64) * 0.5 + 0.5)
    ast_src = np.minimum((par * np.arange(1, n)).astype(np.int64), np.arange(1, n) - 1)
    ast_dst = np.arange(1, n, dtype=np.int64)
    # CFG: chain i -> i+1 with occasional forward/backward branch; last node loops to 0
    br = synth.unit(tag + "/cfgb", n, salt)
    tgt = synth.ints(tag + "/cfgt", (n,), 0, n, salt).numpy()
    cfg_src = np.arange(n, dtype=np.int64)
    cfg_dst = np.where(br > 0.7, tgt, (np.arange(n) + 1) % n).astype(np.int64)
    # CDG: random pairs, duplicates allowed
    cdg_src = synth.ints(tag + "/cdgs", (n,), 0, n, salt).numpy()
    cdg_dst = synth.ints(tag + "/cdgd", (n,), 0, n, salt).numpy()
    loops = np.arange(n, dtype=np.int64)
    src = np.concatenate([ast_src, cfg_src, cdg_src, loops])
    dst = np.concatenate([ast_dst, cfg_dst, cdg_dst, loops])
    etype = np.concatenate([np.zeros(n - 1), np.ones(n), np.full(n, 2), np.zeros(n)]).astype(np.int64)
    node = synth.tensor(tag + "/emb", (n, emb), -0.85, 0.85, salt)
    pos = synth.tensor(tag + "/pos", (n, 4), 0.0, 1.0, salt)
    miss = torch.from_numpy(synth.unit(tag + "/miss", n, salt) > 0.8)
    pos[miss] = 0.0
    g = BatchedGraph(torch.from_numpy(src), torch.from_numpy(dst), [n],
                     {"_UNIX_NODE_EMB": node, "pos_emb": pos,
                      "_lineno": torch.arange(1, n + 1, dtype=torch.int64)},
                     {"_ETYPE": torch.from_numpy(etype)})
    return g


def make_ids(index: int, length=512, vocab=51416, pad=1, lo=128, salt=0) -> torch.Tensor:
    tag = f"ids/{index}"
    valid = int(synth.ints(tag + "/len", (1,), min(lo, length), length + 1, salt)[0])
    ids = synth.ints(tag + "/tok", (length,), 5, vocab, salt)
    ids[0], ids[1], ids[2] = 0, 6, 2
    ids[valid - 1] = 2
    ids[valid:] = pad
    return ids


def make_line_ids(index: int, n_lines: int, length=64, vocab=51416, pad=1, lo=4, salt=0):
    """Token ids of the source LINES of function `index` (one row per graph node: data_list.py:293-299 encodes every line of the
    function on its own, padded to a fixed length), with their non-pad counts: ([n_lines, length] int64, [n_lines] int32).
    Lines are short: lengths ~ U[lo, length]."""
    tag = f"lines/{index}"
    lens = synth.ints(tag + "/len", (n_lines,), lo, length + 1, salt)
    ids = synth.ints(tag + "/tok", (n_lines, length), 5, vocab, salt)
    ids[:, 0], ids[:, 1], ids[:, 2] = 0, 6, 2
    col = torch.arange(length)[None, :]
    ids[col == (lens[:, None] - 1)] = 2
    ids[col >= lens[:, None]] = pad
    return ids, lens.to(torch.int32)


def make_image(index: int, size=448, salt=0) -> torch.Tensor:
    return synth.tensor(f"img/{index}", (3, size, size), -1.7, 1.7, salt)


def make_label(index: int, salt=0) -> int:
    return int(synth.unit(f"label/{index}", 1, salt)[0] > 0)


def make_image_u8(index: int, h: int, w: int, salt=0) -> torch.Tensor:
    """A decoded RGB image as the loader would hand it over: uint8 [h, w, 3] (deterministic per index)."""
    return synth.ints(f"img8/{index}", (h, w, 3), 0, 256, salt).to(torch.uint8)


def make_batch(indices, img_size=448, seq_len=512, vocab=51416, n_lo=150, n_hi=250, salt=0, tok_lo=128):
    """(graph, images [B,3,S,S], ids [B,L], labels [B]) for the fused model.  Non-pad tokens per function ~ U[tok_lo, seq_len]
    (tok_lo = seq_len: every function fills its row)."""
    graphs = [make_graph(i, n_lo, n_hi, salt=salt) for i in indices]
    g = batch_graphs(graphs)
    images = torch.stack([make_image(i, img_size, salt) for i in indices])
    ids = torch.stack([make_ids(i, seq_len, vocab, lo=tok_lo, salt=salt) for i in indices])
    labels = torch.tensor([make_label(i, salt) for i in indices], dtype=torch.int64)
    return g, images, ids, labels


def make_joern_cpg(index: int, n_lines=40, salt=0):
    """A synthetic Joern export of one function: (nodes_json, edges_json) in the layout ``svdj.get_node_edges`` reads
    (sastvd/helpers/joern.py:260-275: node records with id / _label / name / code / lineNumber / controlStructureType, edge rows
    [innode, outnode, etype, dataflow]).  It carries everything the reference's filters act on: META_DATA / FILE / COMMENT / <global>
    records, nodes without a line number (TYPE, METHOD_RETURN), "<empty>" and empty code, several CPG nodes per source line, duplicate
    line-level edges, CONTAINS / DOMINATE / POST_DOMINATE / SOURCE_FILE / REACHING_DEF / EVAL_TYPE / REF edges next to AST / CFG / CDG, lines
    that only non-kept edge types touch.  Every edge leaves a node that has a line (see tests/golden/make_golden.py: the reference's own
    TYPE-pseudo-node loop cannot run on pandas >= 2)."""
    rng = np.random.default_rng(int(synth.name_seed(f"joern/{index}", salt)))
    nodes, edges, nid = [], [], [1000]

    def node(label, name="", code="", line=None, cst=None):
        r = {"id": nid[0], "_label": label, "name": name, "code": code}
        if line is not None:
            r["lineNumber"] = int(line)
        if cst is not None:
            r["controlStructureType"] = cst
        nodes.append(r)
        nid[0] += int(rng.integers(1, 4))
        return r["id"]

    node("META_DATA", "", "<empty>")
    node("FILE", "f.c", "<empty>")
    node("NAMESPACE_BLOCK", "<global>", "<global>", 1)
    types = [node("TYPE", t, "") for t in ("int", "char", "size_t")]
    method = node("METHOD", f"fn{index}", f"int fn{index} (char *p, size_t n)", 1)
    ret = node("METHOD_RETURN", "RET", "RET")
    node("COMMENT", "", "/* c */", 2)
    per_line = {}
    for ln in range(2, n_lines + 2):
        k = int(rng.integers(1, 5))
        ids = []
        for j in range(k):
            kind = int(rng.integers(0, 5))
            ident = "v" * int(rng.integers(1, 9)) + str(int(rng.integers(0, 99)))
            if kind == 0:
                ids.append(node("CALL", "<operator>.assignment", f"{ident} = {ident} + {j}", ln))
            elif kind == 1:
                ids.append(node("IDENTIFIER", ident, ident, ln))
            elif kind == 2:
                ids.append(node("LITERAL", str(j), "<empty>" if j % 2 else "", ln))
            elif kind == 3:
                ids.append(node("CONTROL_STRUCTURE", "", f"if ({ident} < n)", ln, "IF"))
            else:
                ids.append(node("LOCAL", ident, f"int {ident}", ln))
        # a unique longest code per line: pandas picks a line's representative with an unstable sort, ties are not defined by the reference
        used = set()
        for r in nodes[-k:]:
            eff = r["code"] if r["code"] not in ("", "<empty>") else r["name"]
            while len(eff) in used:
                eff += "_"
            used.add(len(eff))
            if r["code"] in ("", "<empty>"):
                r["name"] = eff
            else:
                r["code"] = eff
        per_line[ln] = ids
    lines = sorted(per_line)

    def edge(outn, inn, et, df=""):
        edges.append([inn, outn, et, df])

    for ln in lines:                                        # AST: method -> first node of a line -> the rest of the line
        ids = per_line[ln]
        if rng.random() < 0.85:
            edge(method, ids[0], "AST")
        for a in ids[1:]:
            edge(ids[0], a, "AST")
            edge(ids[0], a, "CONTAINS")
    for a, b in zip(lines[:-1], lines[1:]):                 # CFG chain with some repeats (duplicate line-level edges) and skips
        if rng.random() < 0.8:
            edge(per_line[a][-1], per_line[b][0], "CFG")
            if rng.random() < 0.3:
                edge(per_line[a][0], per_line[b][-1], "CFG")
        edge(per_line[a][0], per_line[b][0], "DOMINATE")
        edge(per_line[b][0], per_line[a][0], "POST_DOMINATE")
    for _ in range(n_lines):                                # CDG / REACHING_DEF / REF between random lines
        a, b = (int(x) for x in rng.choice(lines, 2))
        edge(per_line[a][0], per_line[b][-1], ("CDG", "REACHING_DEF", "REF")[int(rng.integers(0, 3))], "x")
    for ln in lines[::3]:                                   # into nodes without a line: dropped by the line filters
        edge(per_line[ln][0], types[int(rng.integers(0, 3))], "EVAL_TYPE")
        edge(per_line[ln][-1], ret, "CFG")
    edge(method, nodes[1]["id"], "SOURCE_FILE")
    return nodes, edges


def write_corpus(root, splits, img_hw=(96, 128), seq_len=64, vocab=1000, line_len=16, n_lines=24, head_only=False, salt=0):
    """Write a small corpus directory in the reference's FILE FORMATS (the layout data/bigvul_dataset.py:BigVulFiles documents) from
    synthetic content: ``splits`` = {"train": [ids], "val": [...], "test": [...]}.  PNGs of ``img_hw`` (any size: the device transform
    resizes), Joern exports from ``make_joern_cpg``, OCR boxes for most lines, function token ids, per-line token ids keyed by line
    number; with ``head_only`` also the cached encoder features the reference's head-only step reads.  Test / demo helper: the real
    corpus is an external download."""
    import json
    import os
    import pickle
    from PIL import Image
    root = str(root)
    for d in ("images", "func_before", "norm_pos_dict", "token_ids", "line_token_ids", "swinv2_method_level_try5"):
        os.makedirs(os.path.join(root, d), exist_ok=True)
    rows = []
    for split, ids in splits.items():
        with open(os.path.join(root, f"{split}.txt"), "w") as f:
            for _id in ids:
                f.write(f"images/{_id}.png {make_label(_id, salt)}\n")
        for _id in ids:
            Image.fromarray(make_image_u8(_id, *img_hw, salt=salt).numpy(), "RGB").save(os.path.join(root, "images", f"{_id}.png"))
            nodes, edges = make_joern_cpg(_id, n_lines=n_lines, salt=salt)
            with open(os.path.join(root, "func_before", f"{_id}.c.nodes.json"), "w") as f:
                json.dump(nodes, f)
            with open(os.path.join(root, "func_before", f"{_id}.c.edges.json"), "w") as f:
                json.dump(edges, f)
            lines = sorted({int(n["lineNumber"]) for n in nodes if n.get("lineNumber") not in (None, "")})
            rng = np.random.default_rng(int(synth.name_seed(f"corpus/{_id}", salt)))
            pos = {ln: rng.random(4).astype(np.float32).tolist() for ln in lines if rng.random() < 0.8}      # some lines the OCR missed
            with open(os.path.join(root, "norm_pos_dict", f"{_id}.pkl"), "wb") as f:
                pickle.dump(pos, f)
            np.save(os.path.join(root, "token_ids", f"{_id}.npy"), make_ids(_id, seq_len, vocab, lo=seq_len // 4, salt=salt).numpy())
            lids, _ = make_line_ids(_id, len(lines), length=line_len, vocab=vocab, lo=1, salt=salt)
            np.savez(os.path.join(root, "line_token_ids", f"{_id}.npz"), lineno=np.asarray(lines, dtype=np.int64), ids=lids.numpy())
            if head_only:
                torch.save(synth.tensor(f"imgfeat/{_id}", (1024,)), os.path.join(root, "swinv2_method_level_try5", f"{_id}.pt"))
                rows.append((_id, synth.tensor(f"txtfeat/{_id}", (768,)).tolist()))
    if head_only:
        import pandas as pd
        pd.DataFrame({"ids": [r[0] for r in rows], "repr": [r[1] for r in rows]}).to_pickle(os.path.join(root, "result.pkl"))
    return root
