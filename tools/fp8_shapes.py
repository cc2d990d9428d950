#!/usr/bin/env python3
"""Per-shape times of the e4m3 forward products against their bf16 forms (batch-256 inference shapes of both encoders).

    python tools/fp8_shapes.py [--rows 0|128..256]      rows: tile height of the persistent kernel (0 = the host's pick)

The e4m3 full-line ring runs on v_mfma_scale_f32_16x16x128_f8f6f4 up to 192-row tiles and on the non-scaled 16x16x32 form above."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mvuld_amd import hip, ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--reps", type=int, default=10)
args = ap.parse_args()
B = args.batch
dev = torch.device("cuda:0")
shapes = []
for s, (T, C) in enumerate([(B * 3136, 256), (B * 784, 512), (B * 196, 1024)], start=1):
    shapes += [(f"s{s}.qkv", T, 3 * C, C, hip.EPI_BIAS), (f"s{s}.proj", T, C, C, hip.EPI_BIAS), (f"s{s}.fc1", T, 4 * C, C, hip.EPI_GELU), (f"s{s}.fc2", T, C, 4 * C, hip.EPI_BIAS)]
T, C = B * 512, 768
shapes += [("rob.qkv", T, 3 * C, C, hip.EPI_BIAS), ("rob.out", T, C, C, hip.EPI_BIAS), ("rob.fc1", T, 4 * C, C, hip.EPI_GELU), ("rob.fc2", T, C, 4 * C, hip.EPI_BIAS)]


def timed(fn):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / args.reps * 1000


rows = hip.LIB.fn("mvuld_set_gemm_p256_rows")
print(f"{'product':10s} {'M':>7s} {'N':>5s} {'K':>5s} | {'bf16 us':>8s} {'frac':>5s} | " + " ".join(f"{'e4m3@' + str(r):>9s}" for r in ("pick", 192, 224, 256)) + " | best e4m3 / bf16")
for tag, M, N, K, epi in shapes:
    a = (torch.rand((M, K), device=dev) - 0.5).to(torch.bfloat16)
    w = ((torch.rand((N, K), device=dev) - 0.5) * 0.1).to(torch.bfloat16)
    bias = torch.rand((N,), device=dev) - 0.5
    out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    rows(0)
    t16 = timed(lambda: ops.gemm_nt(a, w, bias=bias, epi=epi, out=out))
    qa, sa = ops.quant_fp8(a)
    qw, sw = ops.quant_fp8(w)
    t8 = []
    for r in (0, 192, 224, 256):
        rows(r)
        t8.append(timed(lambda: ops.gemm_nt_fp8(qa, sa, qw, sw, bias=bias, epi=epi, out=out)))
    rows(0)
    print(f"{tag:10s} {M:7d} {N:5d} {K:5d} | {t16:8.1f} {2.0 * M * N * K / t16 / 1e6 / 2.5e3:5.2f} | " + " ".join(f"{t:9.1f}" for t in t8) + f" | {min(t8) / t16:.2f}")
    del a, w, out, qa, qw
