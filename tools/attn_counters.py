#!/usr/bin/env python3
"""Condense tools/attn_counters.sh output into profiles/<tag>_attn_counters.csv: one row per (attention pass, geometry) with the average
launch duration (kernel trace, un-instrumented run) and the SQ counters of the --pmc run, as sums over the chip per launch and as the
ratios the LDS-bound diagnosis rests on (LDS array cycles / CU busy cycles, bank-conflict share of the LDS cycles, MFMA-busy share).

    python tools/attn_counters.py <tag> [gpurun_out/prof/attn_<tag>]"""
import collections, csv, glob, os, re, sys

tag = sys.argv[1]
src = sys.argv[2] if len(sys.argv) > 2 else f"gpurun_out/prof/attn_{tag.split('_')[-1]}"


def short(n):
    m = re.search(r"(attn_\w+?_k)(?:I(.*?)E?Ev|<([^>]*)>)", n)
    if not m or "reduce" in n:
        return None
    if m.group(3) is not None:
        args = [a.strip() for a in m.group(3).split(",")]
    else:           # mangled: ILi32ELi0ELb1E...
        args = [("true" if v == "1" else "false") if t == "b" else v for t, v in re.findall(r"L([ib])(\d+)E", "L" + m.group(2).lstrip("L") + "E")]
    return f"{m.group(1)}<{','.join(args)}>"


def geometry(kernel, grid, wg):
    nwg = grid // max(wg, 1)
    names = {2048: "swin s0 (2048 x N784 h4)", 1024: "swin s1/s3 (1024)", 512: "swin s2 (512 x N784 h16)", 384: "text (384 x N512 hd64)", 768: "text split2 (768)"}
    return names.get(nwg, f"{nwg} workgroups")


dur = collections.defaultdict(lambda: [0, 0])
for p in glob.glob(f"{src}/kt/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        k = short(r["Kernel_Name"])
        if not k:
            continue
        key = (k, int(r["Grid_Size_X"]), int(r["Workgroup_Size_X"]), int(r.get("LDS_Block_Size", 0) or 0))
        dur[key][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        dur[key][1] += 1
cnt = collections.defaultdict(lambda: collections.defaultdict(float))
ncnt = collections.defaultdict(lambda: collections.defaultdict(int))
vg = {}
for p in glob.glob(f"{src}/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        k = short(r["Kernel_Name"])
        if not k:
            continue
        key = (k, int(r["Grid_Size"]), int(r["Workgroup_Size"]), int(r.get("LDS_Block_Size", 0) or 0))
        cnt[key][r["Counter_Name"]] += float(r["Counter_Value"])
        ncnt[key][r["Counter_Name"]] += 1
        vg[key] = (r.get("VGPR_Count", ""), r.get("Accum_VGPR_Count", ""), r.get("SGPR_Count", ""))
C = ["SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_WAIT_INST_LDS", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY",
     "SQ_ACTIVE_INST_VALU"]
os.makedirs("profiles", exist_ok=True)
out = f"profiles/{tag}_attn_counters.csv"
with open(out, "w") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "geometry", "workgroups", "threads", "lds_bytes", "vgpr", "launches", "avg_us"] + [c + "_per_launch" for c in C]
               + ["lds_active/cu_busy", "bank_conflict/lds_active", "mfma_busy/cu_busy(x1/4 SIMD)", "wait_any/wave_cycles", "valu_active/wave_cycles"])
    for key in sorted(dur, key=lambda k: (k[0], -k[1])):
        k, grid, wg, lds = key
        d, n = dur[key]
        c = {x: (cnt[key][x] / ncnt[key][x] if ncnt[key][x] else float("nan")) for x in C}
        busy = c["SQ_BUSY_CU_CYCLES"] or float("nan")
        row = [k, geometry(k, grid, wg), grid // wg, wg, lds, vg.get(key, ("",))[0], n, f"{d / n / 1e3:.1f}"] + [f"{c[x]:.4g}" for x in C]
        row += [f"{c['SQ_LDS_IDX_ACTIVE'] / busy:.3f}", f"{c['SQ_LDS_BANK_CONFLICT'] / (c['SQ_LDS_IDX_ACTIVE'] or float('nan')):.3f}",
                f"{c['SQ_VALU_MFMA_BUSY_CYCLES'] / busy / 4:.3f}", f"{c['SQ_WAIT_ANY'] / (c['SQ_WAVE_CYCLES'] or float('nan')):.3f}",
                f"{c['SQ_ACTIVE_INST_VALU'] / (c['SQ_WAVE_CYCLES'] or float('nan')):.3f}"]
        w.writerow(row)
print(open(out).read())
