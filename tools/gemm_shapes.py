#!/usr/bin/env python3
"""Per-shape table of the step's GEMMs (diagnostic; feeds profiles/rNN_gemm_shapes.csv).

Runs every distinct (M, N, K, epilogue) product of the fused train step at batch 32 -- forward, data-gradient (NT kernel
family) and weight-gradient (TN kernel) -- REPS times each, in a fixed order, and prints live HIP-event timings.  Under
``rocprofv3 --kernel-trace`` / ``--pmc ...`` the same fixed order lets ``tools/summarize_gemm_shapes.py`` attribute the
dispatches of the (shape-agnostic) kernel names back to shapes.

    python tools/gemm_shapes.py [--reps 5] [--only nt|tn] [--csv out.csv]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mvuld_amd import hip, ops  # noqa: E402

B = int(os.environ.get("GS_BATCH", 32))
# "gelu" / "dgelu" rows are issued as the step issues them: the training pair EPI_GELU_DG / EPI_MUL_AUX (GS_GELU_DG=0: the round-2 pair)
_DG = os.environ.get("GS_GELU_DG", "1") != "0"
EPI = {"none": hip.EPI_NONE, "bias": hip.EPI_BIAS, "gelu": hip.EPI_GELU_DG if _DG else hip.EPI_GELU,
       "dgelu": hip.EPI_MUL_AUX if _DG else hip.EPI_MUL_DGELU, "addaux": hip.EPI_ADD_AUX}


GROUPED = os.environ.get("GS_GROUPED", "1") != "0"      # weight gradients of a block as the step issues them: one grouped launch (kind "tg")


def step_shapes():
    """[(tag, kind, M, N, K, epi, launches_per_step)]  kind: nt | tn | tg (a block's weight gradients in one grouped launch: N, K = 0,
    the member products are the preceding `tn` rows of the same stage, listed with 0 launches per step)"""
    out = []
    # SwinV2-base 448: (tokens, C, blocks)
    for s, (T, C, nb) in enumerate([(B * 12544, 128, 2), (B * 3136, 256, 2), (B * 784, 512, 18), (B * 196, 1024, 2)]):
        out += [(f"s{s}.qkv", "nt", T, 3 * C, C, "bias", nb), (f"s{s}.proj", "nt", T, C, C, "bias", nb),
                (f"s{s}.fc1", "nt", T, 4 * C, C, "gelu", nb), (f"s{s}.fc2", "nt", T, C, 4 * C, "bias", nb),
                (f"s{s}.fc2^T", "nt", T, 4 * C, C, "dgelu", nb), (f"s{s}.fc1^T", "nt", T, C, 4 * C, "addaux", nb),
                (f"s{s}.proj^T", "nt", T, C, C, "none", nb), (f"s{s}.qkv^T", "nt", T, C, 3 * C, "addaux", nb),
                (f"s{s}.dWqkv", "tn", T, 3 * C, C, "", 0 if (GROUPED and s > 0) else nb), (f"s{s}.dWproj", "tn", T, C, C, "", 0 if (GROUPED and s > 0) else nb),
                (f"s{s}.dWfc1", "tn", T, 4 * C, C, "", 0 if (GROUPED and s > 0) else nb), (f"s{s}.dWfc2", "tn", T, C, 4 * C, "", 0 if (GROUPED and s > 0) else nb)]
        if GROUPED and s > 0:
            out += [(f"s{s}.dWblock", "tg", T, 0, C, "", nb)]
        if s < 3:
            out += [(f"s{s}.merge", "nt", T // 4, 2 * C, 4 * C, "none", 1), (f"s{s}.merge^T", "nt", T // 4, 4 * C, 2 * C, "none", 1),
                    (f"s{s}.dWmerge", "tn", T // 4, 2 * C, 4 * C, "", 1)]
    T, C, F, nb = B * 512, 768, 3072, 12
    out += [("rob.qkv", "nt", T, 3 * C, C, "bias", nb), ("rob.out", "nt", T, C, C, "bias", nb), ("rob.fc1", "nt", T, F, C, "gelu", nb),
            ("rob.fc2", "nt", T, C, F, "bias", nb), ("rob.fc2^T", "nt", T, F, C, "dgelu", nb), ("rob.fc1^T", "nt", T, C, F, "addaux", nb),
            ("rob.out^T", "nt", T, C, C, "none", nb), ("rob.qkv^T", "nt", T, C, 3 * C, "addaux", nb),
            ("rob.dWqkv", "tn", T, 3 * C, C, "", 0 if GROUPED else nb), ("rob.dWout", "tn", T, C, C, "", 0 if GROUPED else nb),
            ("rob.dWfc1", "tn", T, F, C, "", 0 if GROUPED else nb), ("rob.dWfc2", "tn", T, C, F, "", 0 if GROUPED else nb)]
    if GROUPED:
        out += [("rob.dWblock", "tg", T, 0, C, "", nb)]
    out += [("sq4096", "nt", 4096, 4096, 4096, "none", 0), ("sq4096.tn", "tn", 4096, 4096, 4096, "", 0)]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--only", default="")
    ap.add_argument("--csv", default="")
    ap.add_argument("--filter", default="")
    ap.add_argument("--p256-mode", type=int, default=-1, help="mvuld_set_gemm_p256_mode (0 never, 1 default rule, 2 always)")
    ap.add_argument("--p256-rows", type=int, default=-1, help="mvuld_set_gemm_p256_rows (0 auto, 128..256)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    hip.LIB.load()
    if args.p256_mode >= 0:
        hip.LIB.fn("mvuld_set_gemm_p256_mode")(args.p256_mode)
    if args.p256_rows >= 0:
        hip.LIB.fn("mvuld_set_gemm_p256_rows")(args.p256_rows)
    rows = []
    gen = torch.Generator(device=dev).manual_seed(1)
    for tag, kind, M, N, K, epi, per_step in step_shapes():
        if args.only and kind != args.only and not (args.only == "tn" and kind == "tg"):
            continue
        if args.filter and args.filter not in tag:
            continue
        x = torch.randn(M, K, device=dev, generator=gen).to(torch.bfloat16)
        fl = 2.0 * M * N * K
        if kind == "nt":
            w = (torch.randn(N, K, device=dev, generator=gen) * 0.05).to(torch.bfloat16)
            bias = torch.randn(N, device=dev, generator=gen) if epi in ("bias", "gelu") else None
            y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            aux = None
            if epi == "gelu":
                aux = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            elif epi in ("dgelu", "addaux"):
                aux = torch.randn(M, N, device=dev, generator=gen).to(torch.bfloat16)

            def fn():
                ops.gemm_nt(x, w, out=y, bias=bias, epi=EPI[epi] if epi != "bias" else hip.EPI_BIAS, aux=aux)
            # algorithmic HBM bytes: A + W read once, C written once (+ aux read or written once)
            by = 2.0 * (M * K + N * K + M * N + (M * N if aux is not None else 0))
        elif kind == "tg":
            # the four weight gradients of a transformer block (hidden = 4K), deferred and issued as one grouped launch (ops.wgrad_group)
            C_, F_ = K, 4 * K
            shapes = [(C_, F_), (F_, C_), (C_, C_), (3 * C_, C_)]          # fc2, fc1, proj / out, qkv: (N, K) of each product
            ops_ = []
            for (n_, k_) in shapes:
                dy = torch.randn(M, n_, device=dev, generator=gen).to(torch.bfloat16)
                xx = torch.randn(M, k_, device=dev, generator=gen).to(torch.bfloat16)
                wp = torch.nn.Parameter(torch.zeros(n_, k_, device=dev))
                wp.grad = torch.zeros(n_, k_, device=dev)
                bp = torch.nn.Parameter(torch.zeros(n_, device=dev))
                bp.grad = torch.zeros(n_, device=dev)
                ops_.append((dy, xx, wp, bp))
            fl = sum(2.0 * M * n_ * k_ for n_, k_ in shapes)

            def fn():
                with ops.wgrad_group():
                    for dy, xx, wp, bp in ops_:
                        ops.linear_wgrad(dy, xx, wp, bp)
            by = sum(2.0 * (M * k_ + M * n_) + 4.0 * n_ * k_ for n_, k_ in shapes)
        else:
            dy = torch.randn(M, N, device=dev, generator=gen).to(torch.bfloat16)
            wp = torch.nn.Parameter(torch.zeros(N, K, device=dev))
            wp.grad = torch.zeros(N, K, device=dev)
            bp = torch.nn.Parameter(torch.zeros(N, device=dev))
            bp.grad = torch.zeros(N, device=dev)

            def fn():
                ops.linear_wgrad(dy, x, wp, bp)
            by = 2.0 * (M * K + M * N) + 4.0 * N * K
        fn()
        fn()
        torch.cuda.synchronize()
        evs = []
        for _ in range(args.reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            evs.append((e0, e1))
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) for a, b in evs)
        us = ts[len(ts) // 2] * 1e3
        rows.append((tag, kind, M, N, K, epi, per_step, us, fl / us / 1e6, by / us / 1e3))
        print(f"{tag:12s} {kind} M={M:6d} N={N:4d} K={K:4d} {epi:6s} x{per_step:2d}  {us:8.1f} us  {fl/us/1e6:7.1f} TFLOP/s  "
              f"{by/us/1e3:7.1f} GB/s(alg)", flush=True)
        del x
    nt = [r for r in rows if r[1] == "nt" and r[6] > 0]
    tn = [r for r in rows if r[1] in ("tn", "tg") and r[6] > 0]
    for name, rs in (("NT", nt), ("TN", tn)):
        if rs:
            tot_us = sum(r[7] * r[6] for r in rs)
            tot_fl = sum(r[8] * 1e6 * r[7] * r[6] for r in rs)            # TFLOP/s x us x launches
            print(f"{name} family, weighted by launches per step: {tot_us/1e3:.2f} ms/step, {tot_fl/tot_us/1e6:.1f} TFLOP/s "
                  f"= {tot_fl/tot_us/1e6/2500:.3f} of dense bf16 peak")
    if args.csv:
        with open(args.csv, "w") as f:
            f.write("shape,kind,M,N,K,epilogue,launches_per_step,us,tflops,alg_GBps\n")
            for r in rows:
                f.write(",".join(str(v if not isinstance(v, float) else round(v, 2)) for v in r) + "\n")


if __name__ == "__main__":
    main()
