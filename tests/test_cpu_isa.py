"""Compiled-code audit of the LDS-DMA kernels (CPU: hipcc cross-compiles gfx950, no GPU needed).

An LDS-DMA ring (``global_load_lds`` / ``buffer_load ... lds`` several steps ahead of the compute behind counted ``s_waitcnt vmcnt(N)``)
only works while the compiler does not drain the DMA queue itself.  hipcc does exactly that when it can see an LDS read next to
outstanding LDS-DMA: it puts ``s_waitcnt vmcnt(0)`` in front of the first read of every step, and the ring silently degrades to one
synchronous load per step -- results stay correct, the kernel just runs 20 % slower (the weight-gradient kernel of rounds 1-2,
DESIGN section 9b item 12).  No numerical test can catch that; this one reads the ISA.
"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mvuld_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _isa(src, tmp_path):
    out = tmp_path / (src + ".s")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-Wno-unused-result",
                    "--cuda-device-only", "-S", os.path.join(CSRC, src), "-o", str(out)], check=True, cwd=CSRC, capture_output=True)
    kernels, name, body = {}, None, []
    for line in open(out):
        m = re.match(r"^(_Z\S+):", line)
        if m:
            name, body = m.group(1), []
            kernels[name] = body
            continue
        t = line.strip()
        if name and t and not t.startswith((";", ".")):
            body.append(t.split(";")[0].strip())
    return kernels


def _drains_before_lds_read(ins, window=5):
    """indices of `s_waitcnt ... vmcnt(0)` followed within `window` instructions by an LDS read"""
    return [k for k, l in enumerate(ins) if l.startswith("s_waitcnt") and "vmcnt(0)" in l
            and any(x.startswith("ds_read") for x in ins[k + 1:k + 1 + window])]


def _uses_lds_dma(ins):
    return any(l.startswith("global_load_lds") or (l.startswith("buffer_load") and l.endswith(" lds")) for l in ins)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
@pytest.mark.parametrize("src,pattern", [("gemm_tn256.hip", "gemm_tn256_k"), ("gemm.hip", "gemm_nt_mfma_bf16")])
def test_lds_dma_ring_is_not_drained_in_front_of_lds_reads(tmp_path, src, pattern):
    kernels = {n: b for n, b in _isa(src, tmp_path).items() if pattern in n and _uses_lds_dma(b)}
    assert kernels, f"no LDS-DMA kernel matching {pattern} in {src}"
    for name, ins in kernels.items():
        assert any(l.startswith("s_waitcnt vmcnt(") and "vmcnt(0)" not in l for l in ins), f"{name}: no counted vmcnt wait at all"
        hits = _drains_before_lds_read(ins)
        assert not hits, f"{name}: s_waitcnt vmcnt(0) directly in front of an LDS read at instruction(s) {hits[:4]}: the ring runs synchronously"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_scalar_atomic_tickets_are_not_read_before_they_land(tmp_path):
    """The dynamic tile walk of gemm_nt_bf16_p256 claims tiles with `s_atomic_add ... glc`, whose ticket returns asynchronously (lgkmcnt).
    hipcc copies an inline-asm OUTPUT operand into its variable's register right behind the statement -- before the ticket has landed --
    so the in-flight ticket lives in s101, a register hipcc never allocates on gfx950, and becomes a C++ value only behind an
    `s_waitcnt lgkmcnt(0)`.  This reads the ISA of every instantiation: a returning scalar atomic either
    targets s101 or is waited for by the very next instruction; s101 is read only right behind a full lgkmcnt wait and written only by
    the `s_mov_b32 s101, 1` that feeds the atomic."""
    kernels = {n: b for n, b in _isa("gemm_p256.hip", tmp_path).items() if "gemm_nt_bf16_p256" in n}
    assert kernels
    seen = 0
    for name, ins in kernels.items():
        for k, l in enumerate(ins):
            m = re.match(r"s_atomic_add (s\d+), ", l)
            if m:
                seen += 1
                assert m.group(1) == "s101" or (ins[k + 1].startswith("s_waitcnt") and "lgkmcnt(0)" in ins[k + 1]), f"{name}: {l} / {ins[k + 1]}"
                if m.group(1) == "s101":
                    assert ins[k - 1] == "s_mov_b32 s101, 1", f"{name}: {ins[k - 1]} in front of {l}"
                continue
            if re.search(r"\bs101\b", l):
                if l == "s_mov_b32 s101, 1":
                    assert ins[k + 1].startswith("s_atomic_add s101"), f"{name}: stray write of s101"
                    continue
                assert re.match(r"s_mov_b32 s\d+, s101$", l), f"{name}: unexpected use of s101: {l}"
                assert ins[k - 1].startswith("s_waitcnt") and "lgkmcnt(0)" in ins[k - 1], f"{name}: s101 read without a wait: {ins[k - 3:k + 1]}"
    assert seen >= 3 * 10, seen          # every bf16 full-line instantiation carries the three claim sites
