#!/usr/bin/env python3
"""Join tools/gemm_shapes.py's live timings with a rocprofv3 --pmc pass of the same script into one per-shape table.

    python tools/summarize_gemm_shapes.py <timings.csv> <pmc_dir> <reps_in_pmc_run> <out.csv> [<traffic_dir>]

The kernels are shape-agnostic, so dispatches are attributed to shapes by order: gemm_shapes.py launches exactly
(2 warm-up + reps) GEMM kernels per shape, in the order of its table.  Counters are averaged over the timed launches.
MFMA busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CU_CYCLES * 4 SIMDs) follows the gfx94x MfmaUtil formula
(rocprofv3 ships no gfx950 derived-counter section); `mfma_floor` = algorithmic MFMA cycles (16 per 16x16x32 bf16
instruction, 1024 SIMDs) / (duration x 2.4 GHz) is the same quantity from first principles.
"""
import collections
import csv
import glob
import sys

csv.field_size_limit(1 << 30)


def load_pmc(d):
    files = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)
    per = collections.OrderedDict()
    for f in files:
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if "gemm_" not in n or "reduce" in n:          # the slab reduction follows its tn256 launch: not a GEMM dispatch
                continue
            k = int(r["Dispatch_Id"])
            e = per.setdefault(k, {"name": n.split("(")[0][:48], "ns": int(r["End_Timestamp"]) - int(r["Start_Timestamp"]),
                                   "vgpr": r["VGPR_Count"], "lds": r["LDS_Block_Size"]})
            e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return [per[k] for k in sorted(per)]


def main():
    timings, pmc_dir, reps, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    traffic_dir = sys.argv[5] if len(sys.argv) > 5 else None
    rows = list(csv.DictReader(open(timings)))
    disp = load_pmc(pmc_dir)
    per = reps + 2
    assert len(disp) == per * len(rows), f"{len(disp)} gemm dispatches for {len(rows)} shapes x {per}"
    traf = load_pmc(traffic_dir) if traffic_dir else None
    if traf is not None:
        assert len(traf) == len(disp)
    with open(out, "w") as f:
        w = csv.writer(f)
        w.writerow(["shape", "kind", "M", "N", "K", "epilogue", "launches_per_step", "us", "tflops", "frac_of_2.5PF", "kernel", "vgpr", "lds_bytes",
                    "mfma_busy_frac", "lds_bank_conflict_frac", "wait_inst_lds_frac", "wait_any_frac", "hbm_read_MB", "hbm_write_MB", "alg_MB"])
        for i, r in enumerate(rows):
            ds = disp[i * per + 2:(i + 1) * per]
            avg = lambda k: sum(d.get(k, 0.0) for d in ds) / len(ds)
            busy = avg("SQ_BUSY_CU_CYCLES")
            wc = avg("SQ_WAVE_CYCLES")
            rd = wr = ""
            if traf is not None:
                ts = traf[i * per + 2:(i + 1) * per]
                rd = round(sum(t.get("FETCH_SIZE", 0.0) for t in ts) / len(ts) * 2 * 1024 / 1e6, 1)     # KiB, x2 gfx950 correction
                wr = round(sum(t.get("WRITE_SIZE", 0.0) for t in ts) / len(ts) * 1024 / 1e6, 1)
            M, N, K = int(r["M"]), int(r["N"]), int(r["K"])
            if r["kind"] == "tg":          # a block's four weight gradients (fc2, fc1, proj / out, qkv at width K): operands once + fp32 gradients
                alg = sum(2.0 * (M * k_ + M * n_) + 4.0 * n_ * k_ for n_, k_ in ((K, 4 * K), (4 * K, K), (K, K), (3 * K, K)))
            elif r["kind"] == "nt":
                alg = 2.0 * (M * K + N * K + M * N + (M * N if r["epilogue"] in ("gelu", "dgelu", "addaux") else 0))
            else:
                alg = 2.0 * (M * K + M * N) + 4.0 * N * K
            w.writerow([r["shape"], r["kind"], M, N, K, r["epilogue"], r["launches_per_step"], r["us"], r["tflops"],
                        round(float(r["tflops"]) / 2500, 3), ds[0]["name"], ds[0]["vgpr"], ds[0]["lds"],
                        round(avg("SQ_VALU_MFMA_BUSY_CYCLES") / (busy * 4) if busy else 0, 3),
                        round(avg("SQ_LDS_BANK_CONFLICT") / busy if busy else 0, 4),
                        round(avg("SQ_WAIT_INST_LDS") / wc if wc else 0, 4), round(avg("SQ_WAIT_ANY") / wc if wc else 0, 3), rd, wr,
                        round(alg / 1e6, 1)])
    print(open(out).read())


if __name__ == "__main__":
    main()
