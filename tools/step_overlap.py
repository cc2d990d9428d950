#!/usr/bin/env python3
"""Stream timeline of ONE TIMED train step from a rocprofv3 --kernel-trace CSV of the default (multi-stream) bench command
-> profiles/rNN_step_overlap.txt.  Answers: where does the step's wall time go once the kernels overlap?

    python tools/step_overlap.py <multi-stream kernel_trace.csv> [<one-stream kernel_trace.csv>] [--pick -4]

* picks a TIMED step (default: the 4th optimizer step from the end: the last ones belong to the instrumented / auxiliary legs)
* per queue (= HIP stream): kernels, busy time, first start / last end, idle time inside its own span
* the chip: time with 0 / 1 / 2 / 3+ kernels in flight
* the critical (main) queue: every idle gap > 20 us with the kernel before and after it and what the other queues ran meanwhile
* per kernel family: summed duration in this step next to the same family's duration in the one-stream trace (co-residency inflation)"""
import collections
import csv
import sys

csv.field_size_limit(1 << 30)
args = [a for a in sys.argv[1:] if not a.startswith("--")]
pick = -4
if "--pick" in sys.argv:
    pick = int(sys.argv[sys.argv.index("--pick") + 1])
S = lambda r: int(r["Start_Timestamp"])
E = lambda r: int(r["End_Timestamp"])


def short(n):
    n = n.replace("void ", "")
    for a, b in (("_Z20attn_bwd_fused_win_k", "attn_bwd_fused_win_k"), ("_Z14attn_fwd_win_k", "attn_fwd_win_k"), ("_Z19transpose_batched_k", "transpose_batched_k")):
        if n.startswith(a):
            return b
    return n.split("(")[0].split("<")[0][:40]


def load(path):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=S)
    ad = [i for i, r in enumerate(rows) if "adamw" in r["Kernel_Name"]]
    ad = [i for n, i in enumerate(ad) if n + 1 == len(ad) or ad[n + 1] != i + 1]       # last launch of each optimizer step
    return rows, ad


def step_of(rows, ad, pick):
    i0, i1 = ad[pick - 1], ad[pick]
    return rows[i0 + 1:i1 + 1], E(rows[i0])


rows, ad = load(args[0])
step, t0 = step_of(rows, ad, pick)
t1 = max(map(E, step))
print(f"file {args[0]}: {len(ad)} optimizer steps in the trace, step {pick} picked: wall {(t1 - t0) / 1e6:.2f} ms, {len(step)} kernels, "
      f"sum of kernel durations {sum(E(r) - S(r) for r in step) / 1e6:.2f} ms")
byq = collections.defaultdict(list)
for r in step:
    byq[r["Queue_Id"]].append(r)
order = sorted(byq, key=lambda q: -sum(E(r) - S(r) for r in byq[q]))
names = {}
for n, q in enumerate(order):
    fams = collections.Counter(short(r["Kernel_Name"]) for r in byq[q])
    names[q] = f"Q{n}"
    rs = byq[q]
    busy = sum(E(r) - S(r) for r in rs)
    lo, hi = min(map(S, rs)), max(map(E, rs))
    print(f"  {names[q]} (queue {q}): {len(rs):5d} kernels, busy {busy / 1e6:6.2f} ms, span {(lo - t0) / 1e6:6.2f} .. {(hi - t0) / 1e6:6.2f} ms, "
          f"idle inside span {(hi - lo - busy) / 1e6:6.2f} ms; mostly {', '.join(f'{k} x{v}' for k, v in fams.most_common(4))}")
pts = []
for r in step:
    pts += [(S(r), 1), (E(r), -1)]
pts.sort()
k, last, hist = 0, t0, collections.Counter()
for t, d in pts:
    hist[min(k, 3)] += t - last
    last = t
    k += d
print("  chip: ms with k kernels in flight: " + ", ".join(f"{k}{'+' if k == 3 else ''}: {v / 1e6:.2f}" for k, v in sorted(hist.items())))
main = byq[order[0]]
others = [r for q in order[1:] for r in byq[q]]
gaps = []
for a, b in zip(main, main[1:]):
    if S(b) - E(a) > 20_000:
        gaps.append((S(b) - E(a), a, b))
tot_gap = sum(S(b) - E(a) for a, b in zip(main, main[1:]))
print(f"  main queue {names[order[0]]}: {tot_gap / 1e6:.2f} ms idle between its kernels; {sum(g[0] for g in gaps) / 1e6:.2f} ms of it in {len(gaps)} gaps > 20 us:")
for g, a, b in sorted(gaps, key=lambda x: -x[0])[:14]:
    inside = collections.Counter()
    for r in others:
        ov = min(E(r), S(b)) - max(S(r), E(a))
        if ov > 0:
            inside[names[r["Queue_Id"]] + ":" + short(r["Kernel_Name"])] += ov
    busy_oth = ", ".join(f"{k} {v / 1e3:.0f}us" for k, v in inside.most_common(3)) or "nothing on any queue (host)"
    print(f"    {g / 1e3:7.0f} us at {(E(a) - t0) / 1e6:6.2f} ms  after {short(a['Kernel_Name'])}  before {short(b['Kernel_Name'])}  | meanwhile: {busy_oth}")
fam = collections.Counter()
cnt = collections.Counter()
for r in step:
    fam[short(r["Kernel_Name"])] += E(r) - S(r)
    cnt[short(r["Kernel_Name"])] += 1
ref = collections.Counter()
if len(args) > 1:
    rows1, ad1 = load(args[1])
    step1, _ = step_of(rows1, ad1, pick)
    for r in step1:
        ref[short(r["Kernel_Name"])] += E(r) - S(r)
    print(f"  one-stream trace {args[1]}: step {pick}: sum of kernel durations {sum(ref.values()) / 1e6:.2f} ms")
print("  family: ms in this step (launches) [ms alone, inflation]")
for n, v in fam.most_common(22):
    extra = f"  [{ref[n] / 1e6:6.3f} alone, x{v / ref[n]:.2f}]" if ref.get(n) else ""
    print(f"    {v / 1e6:7.3f} ms {cnt[n]:5d}x  {n}{extra}")
