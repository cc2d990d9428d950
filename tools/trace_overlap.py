#!/usr/bin/env python3
"""Stream-overlap anatomy of one train step from a rocprofv3 --kernel-trace CSV (diagnostic).

    python tools/trace_overlap.py <kernel_trace.csv>

Takes the last full step (between the last two adamw_k dispatches) and prints: wall time, per-queue busy time and span,
the union of busy intervals, how long k kernels ran concurrently, and per kernel family the summed duration."""
import collections
import csv
import sys

csv.field_size_limit(1 << 30)
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ad = [i for i, r in enumerate(rows) if "adamw" in r["Kernel_Name"]]
ad = [i for n, i in enumerate(ad) if n + 1 == len(ad) or ad[n + 1] != i + 1]       # last launch of each optimizer step
pick = int(sys.argv[2]) if len(sys.argv) > 2 else -2                             # step ends at ad[pick + 1]
i0, i1 = ad[pick], ad[pick + 1] if pick + 1 < 0 else ad[-1]
step = rows[i0 + 1:i1 + 1]
S = lambda r: int(r["Start_Timestamp"])
E = lambda r: int(r["End_Timestamp"])
t0, t1 = S(rows[i0]), E(rows[i1])
t0 = E(rows[i0])
print(f"step wall {(t1 - t0) / 1e6:.2f} ms, {len(step)} kernels, sum of durations {sum(E(r) - S(r) for r in step) / 1e6:.2f} ms")
byq = collections.defaultdict(list)
for r in step:
    byq[(r["Queue_Id"], r.get("Stream_Id", ""))].append(r)
for q, rs in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    busy = sum(E(r) - S(r) for r in rs)
    print(f"  queue/stream {q}: {len(rs):5d} kernels, busy {busy / 1e6:6.2f} ms, span {(min(map(S, rs)) - t0) / 1e6:6.2f} .. {(max(map(E, rs)) - t0) / 1e6:6.2f} ms")
pts = []
for r in step:
    pts += [(S(r), 1), (E(r), -1)]
pts.sort()
k, last, hist = 0, pts[0][0], collections.Counter()
for t, d in pts:
    hist[k] += t - last
    last = t
    k += d
print("  concurrency (ms with k kernels in flight):", {k: round(v / 1e6, 2) for k, v in sorted(hist.items())})
fam = collections.Counter()
cnt = collections.Counter()
for r in step:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")[:44]
    fam[n] += E(r) - S(r)
    cnt[n] += 1
for n, v in fam.most_common(24):
    print(f"  {v / 1e6:7.3f} ms {cnt[n]:5d}x  {n}")
