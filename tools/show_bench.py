#!/usr/bin/env python3
"""Print the headline fields of a bench.py output file (the JSON line is the last line that starts with '{')."""
import json
import sys
for f in sys.argv[1:]:
    d = json.loads([l for l in open(f) if l.startswith("{")][-1])
    print(f, "| train", d["ms_per_step"], "ms,", d["value"], d["unit"])
    for k in ("inference", "fp8", "varlen_text", "roofline", "cpu_baseline", "host_enqueue_ms_per_step", "launches_per_step"):
        if k in d:
            print("   ", k, d[k])
