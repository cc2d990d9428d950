"""Joern CPG export -> line-level function graph (the host half).

The reference does this offline with pandas: ``svdj.get_node_edges`` (sastvd/helpers/joern.py:252-352) reads
``<file>.nodes.json`` / ``<file>.edges.json``, ``ne_groupnodes`` (data/data_list.py:319-340) collapses CPG nodes onto source lines,
``feature_extraction`` (:343-376) keeps the AST / CFG / CDG edges (``svdj.rdg(e, "all")``, joern.py:455-480), drops lines no edge
touches (``drop_lone_nodes`` :535-543) and renumbers; ``ImageList.item`` (:265-317) then builds ``dgl.graph((eo, ei))``, attaches the OCR
position of every line and one self-loop per node.  Here the same steps are plain Python over the JSON records (no DataFrame: a function
has a few hundred CPG nodes), the result is a ``BatchedGraph`` whose CSR index is built on the device (``graph_index.hip``).

Semantics kept from the pandas code, including the accidental ones:
* a node's line number counts as present iff the JSON record carries ``lineNumber`` (pandas: NaN -> "" by ``fillna``);
* the representative of a line is the node with the LONGEST ``code`` (``code`` = ``name`` when empty / "<empty>"); the line-level node
  order is descending code length.  pandas sorts with an unstable quicksort there, so the order among equal lengths is not defined
  by the reference; this module breaks ties by original record order (stable);
* edges are de-duplicated on (line_in, line_out, etype) keeping the first, in the order of the edge file;
* edges with a missing line on either side are dropped (the ``isinstance(x, float)`` filters of :335-336), which also makes the
  "TYPE_" pseudo-nodes ``get_node_edges`` synthesises (:322-343, a ``DataFrame.append`` loop that fails on pandas >= 2) irrelevant to
  the result: they only ever sit on edges with a missing ``line_out``;
* self-loops are appended after the real edges, one per node, no de-duplication (``dgl.add_self_loop``)."""
import json

import numpy as np
import torch

from ..graph import BatchedGraph, add_self_loop

ETYPE_MAP = {"AST": 0, "CDG": 1, "REACHING_DEF": 2, "CFG": 3, "EVAL_TYPE": 4, "REF": 5}          # data_list.py:456-463
_DROP_ETYPES = ("CONTAINS", "SOURCE_FILE", "DOMINATE", "POST_DOMINATE")                              # joern.py:303-306
_GTYPES = {   # svdj.rdg (joern.py:455-492)
    "reftype": ("EVAL_TYPE", "REF"), "ast": ("AST",), "pdg": ("REACHING_DEF", "CDG"), "cfg": ("CFG",), "cdg": ("CDG",),
    "cfgcdg": ("CFG", "CDG"), "all": ("CFG", "CDG", "AST"), "other": ("CFG", "CDG", "REACHING_DEF"),
}


def get_node_edges(nodes_json, edges_json):
    """-> (nodes, edges): nodes = list of dicts {id, _label, name, code, lineNumber (int or None)} after the label / name filters;
    edges = list of (innode, outnode, etype, line_in, line_out) with both endpoints among those nodes, at least one of them on a line,
    in edge-file order  (joern.py:252-320; the inputs are the parsed JSON documents)."""
    nodes = []
    for r in nodes_json:
        label, name = r.get("_label", "") or "", r.get("name", "") or ""
        if name == "<global>" or "META" in label or label in ("COMMENT", "FILE"):
            continue
        code = r.get("code", "") or ""
        if code == "<empty>":
            code = ""
        if code == "":
            code = name
        ln = r.get("lineNumber", None)
        nodes.append({"id": r["id"], "_label": label, "name": name, "code": code, "lineNumber": None if ln is None or ln == "" else int(ln)})
    line = {n["id"]: n["lineNumber"] for n in nodes}
    edges = []
    for e in edges_json:
        innode, outnode, etype = e[0], e[1], e[2]
        if etype in _DROP_ETYPES or outnode not in line or innode not in line:
            continue
        lo, li = line[outnode], line[innode]
        if lo is None and li is None:
            continue
        edges.append((innode, outnode, etype, li, lo))
    return nodes, edges


def feature_extraction(nodes_json, edges_json, graph_type="all"):
    """-> (code [n] list of str, lineno [n] list of int, ei [e], eo [e], etypes [e]) exactly as data_list.py:343-376 returns them (minus
    the unused node-type statistics): node k is source line ``lineno[k]``, edge j runs from node ``eo[j]`` to node ``ei[j]``."""
    nodes, edges = get_node_edges(nodes_json, edges_json)
    # ne_groupnodes: longest code per line, descending length (stable), edges onto lines, first of every (in, out, type)
    with_line = [n for n in nodes if n["lineNumber"] is not None]
    order = sorted(range(len(with_line)), key=lambda i: -len(with_line[i]["code"]))
    reps, seen = [], set()
    for i in order:
        ln = with_line[i]["lineNumber"]
        if ln not in seen:
            seen.add(ln)
            reps.append(with_line[i])
    el, dedup = [], set()
    for (_, _, etype, li, lo) in edges:
        key = (li, lo, etype)
        if key in dedup:
            continue
        dedup.add(key)
        if li is None or lo is None:
            continue
        el.append((li, lo, etype))
    keep = _GTYPES[graph_type.split("+")[0]]
    el = [e for e in el if e[2] in keep]
    touched = {e[0] for e in el} | {e[1] for e in el}
    reps = [n for n in reps if n["lineNumber"] in touched]
    index = {n["lineNumber"]: k for k, n in enumerate(reps)}
    return ([n["code"] for n in reps], [n["lineNumber"] for n in reps], [index[e[0]] for e in el], [index[e[1]] for e in el],
            [ETYPE_MAP[e[2]] for e in el])


def load_cpg(path):
    """The two JSON documents Joern wrote next to ``path`` (``<path>.nodes.json`` / ``<path>.edges.json``, joern.py:260-268)."""
    with open(str(path) + ".nodes.json") as f:
        nodes_json = json.load(f)
    with open(str(path) + ".edges.json") as f:
        edges_json = json.load(f)
    return nodes_json, edges_json


def build_function_graph(nodes_json, edges_json, norm_pos_dict=None, graph_type="all"):
    """ImageList.item without the caches (data_list.py:265-317): -> (BatchedGraph, code lines).  ndata: ``_lineno`` f32 [n], ``pos_emb`` f32
    [n, 4] (the OCR box of the line from ``norm_pos_dict``, zeros when the line was not recognised); edata ``_ETYPE`` i64 (0 on the
    self-loops, as dgl fills new edge features).  ``_UNIX_NODE_EMB`` / ``_FUNC_EMB`` come from the text encoder
    (``MyUniXcoder.encode_lines`` on the returned lines, or ``FusedMVulD.forward(node_ids=...)`` on the device)."""
    code, lineno, ei, eo, et = feature_extraction(nodes_json, edges_json, graph_type)
    n = len(lineno)
    pos = np.zeros((n, 4), dtype=np.float32)
    for k, ln in enumerate(lineno):
        if norm_pos_dict and int(ln) in norm_pos_dict:
            pos[k] = np.asarray(norm_pos_dict[int(ln)], dtype=np.float32).reshape(4)
    g = BatchedGraph(torch.tensor(eo, dtype=torch.int64), torch.tensor(ei, dtype=torch.int64), [n],
                     {"_lineno": torch.tensor(lineno, dtype=torch.float32), "pos_emb": torch.from_numpy(pos)},
                     {"_ETYPE": torch.tensor(et, dtype=torch.int64)})
    code = [c.replace("\\t", "").replace("\\n", "") for c in code]                      # :290
    return add_self_loop(g), code
