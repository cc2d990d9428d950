#!/usr/bin/env python3
"""Per-step kernel-time table from a rocprofv3 --kernel-trace --stats run of bench.py (diagnostic; also writes profiles/<tag>_kernel_stats.csv).

    python tools/kernel_stats.py <dir with *_kernel_stats.csv> <steps incl. warm-up and extra steps> [out.csv]"""
import csv
import glob
import sys

d, steps = sys.argv[1], float(sys.argv[2])
f = glob.glob(f"{d}/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print(f"sum of kernel durations: {tot / 1e6 / steps:.2f} ms/step over {steps:g} steps")
out = sys.argv[3] if len(sys.argv) > 3 else None
w = csv.writer(open(out, "w")) if out else None
if w:
    w.writerow(["kernel", "calls_per_step", "ms_per_step", "avg_us", "percent"])
for r in rows[:48]:
    ms = int(r["TotalDurationNs"]) / 1e6 / steps
    print(f"{ms:8.3f} ms/step {int(r['Calls']) / steps:7.1f}x {float(r['AverageNs']) / 1e3:8.1f} us  {r['Name'][:100]}")
    if w:
        w.writerow([r["Name"][:120], round(int(r["Calls"]) / steps, 1), round(ms, 3), round(float(r["AverageNs"]) / 1e3, 2), r["Percentage"]])
