#!/usr/bin/env python3
"""Condense rocprofv3 output under gpurun_out/prof/{kt,fetch,write} into tracked summaries under profiles/.

  kt    : --kernel-trace --stats            -> profiles/<tag>_kernel_stats.csv (top kernels)
  fetch : --pmc FETCH_SIZE --kernel-trace   \
  write : --pmc WRITE_SIZE --kernel-trace   -> profiles/<tag>_hbm_traffic.csv: per kernel, HBM bytes per launch
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of a wide coalesced read stream
(MI355X_MICROARCH.md, HBM section), so reads are doubled."""
import csv, collections, os, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/prof"
os.makedirs("profiles", exist_ok=True)


def short(n):
    n = n.split("(")[0]
    # the NT GEMM family bench.py's roofline names "gemm_nt_mfma_bf16": every bf16-storage matrix-core NT kernel (persistent 256x256,
    # LDS-DMA rings, the generic tile kernel); "<float" = fp32 output = the head's 3-term-split products, kept apart
    if "gemm_nt" in n and "simple" not in n:
        return "gemm_nt_mfma_bf16(split3 fp32)" if "<float" in n else "gemm_nt_mfma_bf16"
    if "gemm_tn256_k" in n or "gemm_tn_mfma" in n:
        return "gemm_tn_wgrad"
    for k in ("gemm_nt_mfma_bf16", "gemm_tn_mfma_bf16", "gemm_nt_simple", "attn_fwd_mfma_k", "attn_bwd_dq_mfma_k", "attn_bwd_dkv_mfma_k",
              "attn_bwd_dbias_mfma_k", "attn_delta_k", "layernorm_fwd_k", "layernorm_bwd_k", "batchnorm_fwd_k", "batchnorm_bwd_k",
              "transpose_k", "colsum_k", "adamw_k", "cpb_bwd_k", "cpb_fwd_k", "gat_", "embed_", "cast_k", "sumsq_k"):
        if k in n:
            return k if not k.endswith("_") else n
    return n[:60]


def kernel_stats(sub, out):
    import glob
    rows = list(csv.DictReader(open(glob.glob(f"{src}/{sub}/**/*kernel_stats.csv", recursive=True)[0])))
    with open(out, "w") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_ms", "avg_us", "percent"])
        for r in rows[:40]:
            w.writerow([r["Name"][:110], r["Calls"], f"{int(r['TotalDurationNs'])/1e6:.3f}", f"{float(r['AverageNs'])/1e3:.2f}", r["Percentage"]])


# kt: MVULD_CONCURRENT=0 (one stream: per-kernel durations comparable with bench.py's live HIP-event timing)
# kt2: default two-stream run (durations of overlapping kernels stretch; the wall clock is what bench.py reports)
kernel_stats("kt", f"profiles/{tag}_kernel_stats.csv")
if os.path.isdir(f"{src}/kt2"):
    kernel_stats("kt2", f"profiles/{tag}_kernel_stats_two_streams.csv")

agg = collections.defaultdict(lambda: {"n": 0, "fetch": 0.0, "write": 0.0, "ns": 0})
for kind in ("fetch", "write"):
    import glob
    ps = glob.glob(f"{src}/{kind}/**/*counter_collection.csv", recursive=True)
    if not ps:
        continue
    p = ps[0]
    for r in csv.DictReader(open(p)):
        k = short(r["Kernel_Name"])
        a = agg[k]
        a[kind] += float(r["Counter_Value"])
        if kind == "fetch":
            a["n"] += 1
            a["ns"] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
with open(f"profiles/{tag}_hbm_traffic.csv", "w") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "launches", "read_MB_per_launch(x2 corrected)", "write_MB_per_launch", "total_GB_per_pass"])
    for k, a in sorted(agg.items(), key=lambda kv: -(kv[1]["fetch"] * 2 + kv[1]["write"])):
        if a["n"] == 0:
            continue
        rd, wr = a["fetch"] * 2 * 1024 / a["n"], a["write"] * 1024 / a["n"]
        w.writerow([k, a["n"], f"{rd/1e6:.2f}", f"{wr/1e6:.2f}", f"{(a['fetch']*2+a['write'])*1024/1e9:.2f}"])
print(open(f"profiles/{tag}_kernel_stats.csv").read()[:1500])
print(open(f"profiles/{tag}_hbm_traffic.csv").read()[:1500])
