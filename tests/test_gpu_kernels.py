"""GPU parity of individual C-ABI kernels against plain PyTorch fp32 on the CPU (sizes: seconds)."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from util import rel, tol, synth

pytestmark = pytest.mark.gpu

DTYPES = [torch.float32, torch.bfloat16]


def T(name, shape, lo=-1.0, hi=1.0):
    return synth.tensor("kt/" + name, shape, lo, hi)


def dev(x, dtype=None, d="cuda:0"):
    x = x.to(d)
    return x.to(dtype) if dtype is not None else x


def rt(x, dtype):
    """round-trip through the storage dtype (so the CPU reference sees the same inputs)."""
    return x.to(dtype).float()


# ------------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", [(70, 50, 36), (256, 384, 128), (300, 200, 72), (129, 131, 64), (64, 2, 1536)])
def test_gemm_nt_bias_epilogues(gpu, dtype, M, N, K):
    from mvuld_amd import ops, hip
    a, b = rt(T("ga", (M, K)), dtype), rt(T("gb", (N, K)), dtype)
    bias = T("gbias", (N,))
    ref = a @ b.t() + bias
    A, B_, Bi = dev(a, dtype), dev(b, dtype), dev(bias)
    out = ops.gemm_nt(A, B_, bias=Bi)
    assert rel(out, ref) < tol(dtype)
    aux = torch.empty((M, N), dtype=dtype, device=gpu)
    out = ops.gemm_nt(A, B_, bias=Bi, epi=hip.EPI_GELU, aux=aux)
    assert rel(aux, ref) < tol(dtype)
    assert rel(out, F.gelu(ref)) < tol(dtype)
    out = ops.gemm_nt(A, B_, bias=Bi, epi=hip.EPI_ELU)
    assert rel(out, F.elu(ref)) < tol(dtype)
    pre = rt(T("gpre", (M, N), -2, 2), dtype)
    out = ops.gemm_nt(A, B_, epi=hip.EPI_MUL_DGELU, aux=dev(pre, dtype))
    p = pre.clone().requires_grad_(True)
    F.gelu(p).sum().backward()
    assert rel(out, (a @ b.t()) * p.grad) < tol(dtype)
    out = ops.gemm_nt(A, B_, epi=hip.EPI_ADD_AUX, aux=dev(pre, dtype))
    assert rel(out, a @ b.t() + pre) < tol(dtype)
    # the FFN's training pair: GELU whose side output is gelu'(pre-activation), and the product that multiplies by it
    out = ops.gemm_nt(A, B_, bias=Bi, epi=hip.EPI_GELU_DG, aux=aux)
    r = ref.clone().requires_grad_(True)
    F.gelu(r).sum().backward()
    assert rel(out, F.gelu(ref)) < tol(dtype) and rel(aux, r.grad) < tol(dtype)
    out = ops.gemm_nt(A, B_, epi=hip.EPI_MUL_AUX, aux=dev(pre, dtype))
    assert rel(out, (a @ b.t()) * pre) < tol(dtype)
    out32 = ops.gemm_nt(A, B_, bias=Bi, out_dtype=torch.float32)
    assert out32.dtype == torch.float32 and rel(out32, ref) < (2e-4 if dtype == torch.float32 else 1e-2)


@pytest.mark.parametrize("M,N,K,batch", [(3200, 512, 512, 1), (100, 512, 100, 32), (100, 100, 512, 32), (70, 65, 36, 1), (130, 64, 33, 3), (64, 200, 7, 1)])
def test_gemm_nt_fp32_split_in_registers(gpu, M, N, K, batch):
    """mvuld_gemm_nt_f32x3: fp32 operands split into bf16 hi / lo parts on the way to LDS, three products per 32-deep step (the head's
    fp32 tail in the bf16 activation mode; rounds 1-2 ran two split launches and a 3K-deep bf16 product).  Against the fp64 product
    (error of the scheme ~2^-16 per term) and against the three-launch path on the same operands; bias, ELU, residual-join and
    accumulate forms; batched operands with their own strides; K not a multiple of 4 / 32, ragged M / N."""
    from mvuld_amd import ops, hip
    g_ = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(batch, M, K, generator=g_)
    b = torch.randn(batch, N, K, generator=g_)
    bias, auxv = torch.randn(N, generator=g_), torch.randn(batch, M, N, generator=g_)
    ref = (a.double() @ b.double().transpose(1, 2))
    A, B_, Bi, AX = a.to(gpu), b.to(gpu), bias.to(gpu), auxv.to(gpu)
    kw = dict(M=M, N=N, K=K, lda=K, ldb=K, ldc=N, batch=batch, sa=M * K, sb=N * K, sc=M * N)
    ops.USE_SPLIT3[0] = True
    try:
        assert ops.SPLIT3_IN_REGISTERS[0]
        out = ops.gemm_nt(A, B_, out=torch.empty(batch, M, N, device=gpu), **kw)
        out0 = out
        assert rel(out, ref.float()) < 3e-5, rel(out, ref.float())
        ops.SPLIT3_IN_REGISTERS[0] = False
        try:
            old = ops.gemm_nt(A, B_, out=torch.empty(batch, M, N, device=gpu), **kw)
        finally:
            ops.SPLIT3_IN_REGISTERS[0] = True
        assert rel(out, old) < 2e-6, rel(out, old)
        out = ops.gemm_nt(A, B_, out=torch.empty(batch, M, N, device=gpu), bias=Bi, **kw)
        assert rel(out, (ref + bias.double()).float()) < 3e-5
        out = ops.gemm_nt(A, B_, out=torch.empty(batch, M, N, device=gpu), bias=Bi, epi=hip.EPI_ELU, **kw)
        assert rel(out, F.elu((ref + bias.double()).float())) < 3e-5
        out = ops.gemm_nt(A, B_, out=torch.empty(batch, M, N, device=gpu), epi=hip.EPI_ADD_AUX, aux=AX, ldaux=N, saux=M * N, **kw)
        assert rel(out, (ref + auxv.double()).float()) < 3e-5
        acc = AX.clone()
        ops.gemm_nt(A, B_, out=acc, out_mode=hip.OUT_ACCUM, alpha=0.5, **kw)
        assert rel(acc, (0.5 * ref + auxv.double()).float()) < 3e-5
        # operands stored transposed ([K, M] / [K, N]): the same product without a transpose pass, in all four combinations
        AT, BT = A.transpose(1, 2).contiguous(), B_.transpose(1, 2).contiguous()
        for ta, tb in ((True, False), (False, True), (True, True)):
            kt = dict(kw, lda=M if ta else K, ldb=N if tb else K, sa=M * K, sb=N * K)
            o2 = ops.gemm_nt(AT if ta else A, BT if tb else B_, out=torch.empty(batch, M, N, device=gpu), ta=ta, tb=tb, **kt)
            assert rel(o2, out0) < 2e-6, (ta, tb, rel(o2, out0))
        # contraction splits adding into C with atomics (the weight-gradient form dW += dY^T X: both operands stored [K, .])
        if batch == 1:
            accw = AX[0].clone()
            ops.gemm_nt(AT[0], BT[0], out=accw, M=M, N=N, K=K, lda=M, ldb=N, ldc=N, ta=True, tb=True, out_mode=hip.OUT_ATOMIC, splitk=3)
            assert rel(accw, (ref[0] + auxv[0].double()).float()) < 3e-5
    finally:
        ops.USE_SPLIT3[0] = False


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_splitk_atomic_and_batched(gpu, dtype):
    from mvuld_amd import ops, hip
    M, N, K = 96, 160, 4096
    a, b = rt(T("sa", (M, K)), dtype), rt(T("sb", (N, K)), dtype)
    C = torch.ones((M, N), dtype=torch.float32, device=gpu)
    ops.gemm_nt(dev(a, dtype), dev(b, dtype), out=C, out_mode=hip.OUT_ATOMIC, splitk=8)
    assert rel(C, a @ b.t() + 1.0) < (2e-4 if dtype == torch.float32 else 5e-3)
    Bn, m, n, k = 5, 100, 100, 512
    a, b = rt(T("ba", (Bn, m, k)), dtype), rt(T("bb", (Bn, n, k)), dtype)
    out = ops.gemm_nt(dev(a, dtype).view(Bn * m, k), dev(b, dtype).view(Bn * n, k), M=m, N=n, K=k, lda=k, ldb=k, batch=Bn,
                      sa=m * k, sb=n * k, alpha=0.01)
    assert rel(out, 0.01 * a @ b.transpose(1, 2)) < tol(dtype)


def test_gemm_mfma_matches_simple(gpu):
    """A=I with an asymmetric B catches fragment-layout transposes; then random data, both paths."""
    from mvuld_amd import ops
    M = N = K = 256
    eye = torch.eye(M, dtype=torch.bfloat16, device=gpu)
    b = (torch.arange(N * K, device=gpu).view(N, K) % 251).to(torch.bfloat16)
    out = ops.gemm_nt(eye, b, out_dtype=torch.float32)
    assert torch.equal(out, b.float().t().contiguous())
    a, b2 = dev(T("ma", (384, 320)), torch.bfloat16), dev(T("mb", (200, 320)), torch.bfloat16)
    fast = ops.gemm_nt(a, b2, out_dtype=torch.float32)
    ops.FORCE_SIMPLE_GEMM[0] = True
    try:
        slow = ops.gemm_nt(a, b2, out_dtype=torch.float32)
    finally:
        ops.FORCE_SIMPLE_GEMM[0] = False
    assert rel(fast, slow) < 1e-5


@pytest.mark.parametrize("K", [520, 544])
def test_gemm_large_ragged_epilogues(gpu, K):
    """A grid of several hundred tiles with ragged M / N / K tails (more workgroups than the chip holds at once, XCD-swizzled
    tile order), every fused epilogue and both output dtypes, against the product of the bf16-rounded operands in fp32."""
    _gemm_large_ragged(gpu, K)


def _gemm_large_ragged(gpu, K):
    from mvuld_amd import ops, hip
    M, N = 6401, 2056                         # K = 544 (a multiple of 32) is eligible for the 256 x 256 LDS-DMA ring kernel
    g = torch.Generator().manual_seed(7)
    a = (torch.rand((M, K), generator=g) - 0.5).to(torch.bfloat16)
    b = (torch.rand((N, K), generator=g) - 0.5).to(torch.bfloat16)
    bias = torch.rand((N,), generator=g) - 0.5
    pre = ((torch.rand((M, N), generator=g) - 0.5) * 4).to(torch.bfloat16)
    A, B_, Bi, P = a.to(gpu), b.to(gpu), bias.to(gpu), pre.to(gpu)
    ref = (A.float() @ B_.float().t()).cpu()
    out = ops.gemm_nt(A, B_, bias=Bi)
    assert rel(out, ref + bias) < 1e-2
    out32 = ops.gemm_nt(A, B_, bias=Bi, out_dtype=torch.float32)
    assert rel(out32, ref + bias) < 2e-5
    aux = torch.empty((M, N), dtype=torch.bfloat16, device=gpu)
    out = ops.gemm_nt(A, B_, bias=Bi, epi=hip.EPI_GELU, aux=aux)
    assert rel(aux, ref + bias) < 1e-2 and rel(out, F.gelu(ref + bias)) < 1e-2
    out = ops.gemm_nt(A, B_, epi=hip.EPI_ADD_AUX, aux=P)
    assert rel(out, ref + pre.float()) < 1e-2
    p = pre.float().clone().requires_grad_(True)
    F.gelu(p).sum().backward()
    out = ops.gemm_nt(A, B_, epi=hip.EPI_MUL_DGELU, aux=P)
    assert rel(out, ref * p.grad) < 1e-2
    out = ops.gemm_nt(A, B_, epi=hip.EPI_MUL_AUX, aux=P)
    assert rel(out, ref * pre.float()) < 1e-2
    r = (ref + bias).clone().requires_grad_(True)
    F.gelu(r).sum().backward()
    out = ops.gemm_nt(A, B_, bias=Bi, epi=hip.EPI_GELU_DG, aux=aux)
    assert rel(aux, r.grad) < 1e-2 and rel(out, F.gelu(ref + bias)) < 1e-2


DYN_DEFAULT = int(os.environ.get("MVULD_GEMM_DYNAMIC_TILES", "0"))      # keep in step with P256_DYNAMIC_DEFAULT of csrc/gemm_p256.hip
K64_DEFAULT = int(os.environ.get("MVULD_P256_K64", "1"))      # keep in step with P256_K64_DEFAULT of csrc/gemm_p256.hip


@pytest.mark.parametrize("M,N,K,rows", [(6401, 2056, 544, 0), (6401, 2056, 544, 128), (6401, 2056, 544, 160), (6401, 2056, 544, 192),
                                         (6401, 2056, 544, 224), (6401, 2056, 544, 256), (20000, 1288, 160, 0), (20000, 1288, 160, 224),
                                         (70000, 512, 128, 0), (70000, 512, 128, 160), (769, 520, 1024, 0),
                                         (6401, 2056, 576, 128), (6401, 2056, 576, 160), (6401, 2056, 576, 192), (6401, 2056, 576, 224),
                                         (6401, 2056, 576, 256), (20000, 1288, 192, 0), (25088, 512, 2048, 0), (16384, 2304, 768, 0),
                                         (80001, 128, 512, 256), (80001, 128, 512, 0), (40000, 384, 128, 256), (33000, 192, 256, 224),
                                         # two ring steps per tile on 160-row tiles, three column tiles: a three-stage ring would refill the bias slice of
                                         # a tile two places ahead of its epilogue (round 4: such launches take the two-stage ring)
                                         (70000, 768, 128, 160), (70000, 1288, 128, 128)])
def test_gemm_p256_persistent_ragged(gpu, M, N, K, rows):
    """The persistent 256 x 256-tile kernel (csrc/gemm_p256.hip) forced on ragged shapes: M and N tails inside the last
    tiles, fewer tiles than CUs / several tiles per workgroup (the LDS-DMA ring and the bias slices run across tile
    boundaries), K from 4 to 32 ring steps, every fused epilogue (register-only epilogue with v_permlane16_swap: a wrong lane
    map shows as permuted 4-column groups) -- against the fp32 product of the bf16-rounded operands."""
    from mvuld_amd import ops, hip
    g = torch.Generator().manual_seed(11)
    a = (torch.rand((M, K), generator=g) - 0.5).to(torch.bfloat16)
    b = (torch.rand((N, K), generator=g) - 0.5).to(torch.bfloat16)
    bias = torch.rand((N,), generator=g) - 0.5
    pre = ((torch.rand((M, N), generator=g) - 0.5) * 4).to(torch.bfloat16)
    A, B_, Bi, P = a.to(gpu), b.to(gpu), bias.to(gpu), pre.to(gpu)
    ref = (A.float() @ B_.float().t()).cpu()
    hip.LIB.fn("mvuld_set_gemm_p256_mode")(2)
    hip.LIB.fn("mvuld_set_gemm_p256_rows")(rows)       # tile height: 0 = chosen per shape, else forced
    pp = hip.LIB.fn("mvuld_set_gemm_p256_pingpong")
    k64 = hip.LIB.fn("mvuld_set_gemm_p256_k64")
    dyn = hip.LIB.fn("mvuld_set_gemm_dynamic_tiles")
    k64(0)

    def both(**kw):
        """The product under the ping-pong schedule (default) and under the lockstep one: same contraction order, so the bf16
        matrices must be EQUAL; a mis-placed barrier or wait of either schedule reads a ring stage early and shows here or in the
        repeated launches below."""
        pp(1)
        o1 = ops.gemm_nt(A, B_, **kw)
        a1 = kw["aux"].clone() if kw.get("epi") in (hip.EPI_GELU, hip.EPI_GELU_DG) else None
        pp(0)
        o0 = ops.gemm_nt(A, B_, **kw)
        pp(1)
        assert torch.equal(o0, o1)
        if a1 is not None:
            assert torch.equal(a1, kw["aux"])
        if K % 64 == 0:
            # 64-deep full-line ring stages (two or three stages, two 32-deep sub-steps each): the same contraction order again
            k64(1)
            for _ in range(2):
                o2 = ops.gemm_nt(A, B_, **kw)
                assert torch.equal(o2, o1)
                if a1 is not None:
                    assert torch.equal(a1, kw["aux"])
            # the dynamic tile walk (tiles after a workgroup's first one claimed from per-XCD counters): same tiles, same arithmetic per
            # tile (mode 2: the first tile claimed too); a launch would start from non-zero counters if the last workgroup of the one before it had not restored them
            for mode in (1, 2, 1):
                dyn(mode)
                for _ in range(2):
                    o3 = ops.gemm_nt(A, B_, **kw)
                    assert torch.equal(o3, o1)
                    if a1 is not None:
                        assert torch.equal(a1, kw["aux"])
            dyn(0)
            k64(0)
        return o1
    try:
        out = both()
        assert rel(out, ref) < 1e-2
        # exact structure check: identical operands through the older kernels must give the same bf16 matrix up to one rounding
        hip.LIB.fn("mvuld_set_gemm_p256_mode")(0)
        old = ops.gemm_nt(A, B_)
        hip.LIB.fn("mvuld_set_gemm_p256_mode")(2)
        assert float((out.float() - old.float()).abs().max()) <= 2.0 ** -7 * float(ref.abs().max())
        out = both(bias=Bi)
        assert rel(out, ref + bias) < 1e-2
        aux = torch.empty((M, N), dtype=torch.bfloat16, device=gpu)
        out = both(bias=Bi, epi=hip.EPI_GELU, aux=aux)
        assert rel(aux, ref + bias) < 1e-2 and rel(out, F.gelu(ref + bias)) < 1e-2
        for _ in range(4):
            again = ops.gemm_nt(A, B_, bias=Bi, epi=hip.EPI_GELU, aux=torch.empty_like(aux))
            assert torch.equal(again, out)
        out = both(epi=hip.EPI_ADD_AUX, aux=P)
        assert rel(out, ref + pre.float()) < 1e-2
        p = pre.float().clone().requires_grad_(True)
        F.gelu(p).sum().backward()
        out = both(epi=hip.EPI_MUL_DGELU, aux=P)          # (not this kernel's any more: the rings of gemm.hip)
        assert rel(out, ref * p.grad) < 1e-2
        # the training pair: same activation as EPI_GELU, gelu' as the side output; then the one-multiply backward epilogue
        act = ops.gemm_nt(A, B_, bias=Bi, epi=hip.EPI_GELU, aux=torch.empty_like(aux))
        dg = torch.empty_like(aux)
        out = both(bias=Bi, epi=hip.EPI_GELU_DG, aux=dg)
        assert torch.equal(out, act)
        r = (ref + bias).clone().requires_grad_(True)
        F.gelu(r).sum().backward()
        assert rel(dg, r.grad) < 1e-2
        out = both(epi=hip.EPI_MUL_AUX, aux=P)
        assert rel(out, ref * pre.float()) < 1e-2
    finally:
        hip.LIB.fn("mvuld_set_gemm_p256_mode")(1)
        hip.LIB.fn("mvuld_set_gemm_p256_rows")(0)
        pp(1)
        k64(K64_DEFAULT)
        dyn(DYN_DEFAULT)


def test_gemm_p256_dynamic_walk_beside_another_grid(gpu):
    """The dynamic tile walk where it matters: two persistent grids on two streams at once, so that the workgroups of the launch that
    comes second start late (their CUs are held: 160 KB of LDS per workgroup) and find the counters partly or wholly drained by the
    workgroups that did start.  Every product must equal the one the static walk gave alone on the chip, launch after launch (a ticket
    handed out twice or not at all is a tile written twice / never: the outputs are poisoned in between), and each stream has its own
    counter block (shared tickets would skip tiles)."""
    from mvuld_amd import ops, hip
    dyn = hip.LIB.fn("mvuld_set_gemm_dynamic_tiles")
    g = torch.Generator().manual_seed(5)
    shapes = [(25088, 1536, 512), (25088, 2048, 512), (25088, 512, 2048), (16384, 2304, 768), (100352, 384, 192), (6272, 4096, 1024)]
    ops_ = []
    for M, N, K in shapes:
        a = (torch.rand((M, K), generator=g) - 0.5).to(torch.bfloat16).to(gpu)
        b = (torch.rand((N, K), generator=g) - 0.5).to(torch.bfloat16).to(gpu)
        bias = (torch.rand((N,), generator=g) - 0.5).to(gpu)
        ops_.append((a, b, bias))
    dyn(0)
    want = [ops.gemm_nt(a, b, bias=bias) for a, b, bias in ops_]
    wantg = [ops.gemm_nt(a, b, bias=bias, epi=hip.EPI_GELU, aux=torch.empty((a.shape[0], b.shape[0]), dtype=torch.bfloat16, device=gpu)) for a, b, bias in ops_]
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    try:
        for rep in range(4):
            dyn(1 + rep % 2)
            got1, got2 = [], []
            with torch.cuda.stream(s1):
                for a, b, bias in ops_:
                    got1.append(ops.gemm_nt(a, b, bias=bias))
            with torch.cuda.stream(s2):
                for a, b, bias in reversed(ops_):
                    got2.append(ops.gemm_nt(a, b, bias=bias, epi=hip.EPI_GELU, aux=torch.empty((a.shape[0], b.shape[0]), dtype=torch.bfloat16, device=gpu)))
            torch.cuda.synchronize()
            for w, o in zip(want, got1):
                assert torch.equal(w, o)
                o.fill_(float("nan"))
            for w, o in zip(reversed(wantg), got2):
                assert torch.equal(w, o)
                o.fill_(float("nan"))
    finally:
        dyn(DYN_DEFAULT)


@pytest.mark.parametrize("dtype,C", [(torch.bfloat16, 768), (torch.bfloat16, 520), (torch.bfloat16, 100), (torch.float32, 768)])
def test_segment_mean_ragged(gpu, dtype, C):
    """Sentence vector over packed rows / dgl.mean_nodes: per-segment mean of ragged row runs (lengths 1 .. 512), the 16-byte row-parallel
    kernel (bf16, C % 8 == 0) and the scalar one, forward and backward, against torch."""
    from mvuld_amd.hip import call, ptr, dt
    lens = [1, 512, 7, 130, 64, 9, 300, 2]
    off = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32)
    T, B = int(off[-1]), len(lens)
    g = torch.Generator().manual_seed(3)
    x = (torch.rand((T, C), generator=g) - 0.5).to(dtype)
    X, O = x.to(gpu), off.to(gpu)
    out = torch.empty((B, C), dtype=dtype, device=gpu)
    call("segment_mean_fwd", ptr(X), ptr(O), ptr(out), B, C, dt(X))
    ref = torch.stack([x[off[b]:off[b + 1]].float().mean(0) for b in range(B)])
    assert float((out.float().cpu() - ref).abs().max()) < (1e-6 if dtype == torch.float32 else 2.5e-3)
    dout = (torch.rand((B, C), generator=g) - 0.5).to(dtype)
    dx = torch.empty((T, C), dtype=dtype, device=gpu)
    call("segment_mean_bwd", ptr(dout.to(gpu)), ptr(O), ptr(dx), B, C, dt(dx))
    refd = torch.cat([(dout[b].float() / lens[b]).expand(lens[b], C) for b in range(B)])
    assert float((dx.float().cpu() - refd).abs().max()) < (1e-6 if dtype == torch.float32 else 2.5e-3)


def _e4m3_deq(q):
    return q.cpu().view(torch.float8_e4m3fn).float()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_quant_e4m3_matches_torch_float8(gpu, dtype):
    """mvuld_quant_e4m3 (per-tensor scale = max|x| / 448, v_cvt_pk_fp8_f32 round-to-nearest-even) against torch's own
    float8_e4m3fn conversion of x / scale on the host: the same codes byte for byte, bar products that land within one fp32 ulp of
    a rounding boundary (x * (1 / scale) on the device vs the host's arithmetic)."""
    from mvuld_amd import ops
    g = torch.Generator().manual_seed(5)
    for n, amp in ((8, 3.0), (4096 * 33 + 8, 0.02), (1 << 21, 700.0)):
        x = (torch.randn((n,), generator=g) * amp).to(dtype)
        x[n // 2] = 0.0
        q, sc = ops.quant_fp8(x.to(gpu))
        scale = float(x.float().abs().max()) / 448.0
        assert abs(float(sc) - scale) <= 1e-6 * scale
        want = (x.float() * (1.0 / sc.cpu())).to(torch.float8_e4m3fn)
        got = q.cpu().view(torch.float8_e4m3fn)
        diff = got.float() != want.float()
        assert float(diff.float().mean()) < 1e-4, float(diff.float().mean())
        assert float((got.float() - want.float()).abs().max()) <= 32.0          # at most one code apart (largest e4m3 spacing)
        assert float(got.float().abs().max()) == 448.0 and float(got.float()[n // 2]) == 0.0
    z, sz = ops.quant_fp8(torch.zeros(64, dtype=dtype, device=gpu))          # all-zero tensor: finite scale, zero codes
    assert float(sz) > 0 and int(z.sum()) == 0


@pytest.mark.parametrize("M,N,K", [(6401, 2056, 576), (20000, 1288, 256), (769, 520, 1024), (25088, 768, 3072), (256, 8, 256)])
def test_gemm_nt_fp8_vs_dequantised_product(gpu, M, N, K):
    """The fp8 (OCP e4m3; v_mfma_scale_f32_16x16x128_f8f6f4 on the K >= 1024 cases and on every 128-deep case whose tile the host picks
    at <= 192 rows, v_mfma_f32_16x16x32_fp8_fp8 on the others) variant of the persistent 256 x 256 kernel: the product of the DEQUANTISED
    operands in fp32 is what the kernel must return (fp32 accumulation: only the bf16 rounding of the output separates them), with
    ragged M / N tails, bias, and the GELU epilogue with its pre-activation side output."""
    from mvuld_amd import ops, hip
    g = torch.Generator().manual_seed(13)
    a = ((torch.rand((M, K), generator=g) - 0.5) * 3).to(torch.bfloat16)
    b = ((torch.rand((N, K), generator=g) - 0.5) * 0.2).to(torch.bfloat16)
    bias = torch.rand((N,), generator=g) - 0.5
    qa, sa = ops.quant_fp8(a.to(gpu))
    qb, sb = ops.quant_fp8(b.to(gpu))
    ref = (_e4m3_deq(qa).to(gpu) @ _e4m3_deq(qb).to(gpu).t()).cpu() * float(sa) * float(sb)
    out = ops.gemm_nt_fp8(qa, sa, qb, sb)
    assert out.dtype == torch.bfloat16 and rel(out, ref) < 4e-3
    assert float((out.float().cpu() - ref).abs().max()) <= 2.0 ** -8 * float(ref.abs().max()) + 1e-6
    Bi = bias.to(gpu)
    out = ops.gemm_nt_fp8(qa, sa, qb, sb, bias=Bi)
    assert rel(out, ref + bias) < 4e-3
    aux = torch.empty((M, N), dtype=torch.bfloat16, device=gpu)
    out = ops.gemm_nt_fp8(qa, sa, qb, sb, bias=Bi, epi=hip.EPI_GELU, aux=aux)
    assert rel(aux, ref + bias) < 4e-3 and rel(out, F.gelu(ref + bias)) < 1e-2
    dg = torch.empty_like(aux)                           # training form: gelu'(pre-activation) as the side output
    r = (ref + bias).clone().requires_grad_(True)
    F.gelu(r).sum().backward()
    assert torch.equal(ops.gemm_nt_fp8(qa, sa, qb, sb, bias=Bi, epi=hip.EPI_GELU_DG, aux=dg), out) and rel(dg, r.grad) < 4e-3
    # the lockstep schedule of the main loop gives the same bytes as the ping-pong one (default), launch after launch
    # (K % 128 == 0: the default is the 128-byte full-line ring; otherwise the 64-byte ring under the ping-pong schedule)
    pp, k64 = hip.LIB.fn("mvuld_set_gemm_p256_pingpong"), hip.LIB.fn("mvuld_set_gemm_p256_k64")
    try:
        for sched in ((0, 1), (0, 0)):                   # (k64, pingpong)
            k64(sched[0])
            pp(sched[1])
            aux0 = torch.empty_like(aux)
            assert torch.equal(ops.gemm_nt_fp8(qa, sa, qb, sb, bias=Bi, epi=hip.EPI_GELU, aux=aux0), out) and torch.equal(aux0, aux)
    finally:
        pp(1)
        k64(K64_DEFAULT)
    for _ in range(3):
        assert torch.equal(ops.gemm_nt_fp8(qa, sa, qb, sb, bias=Bi, epi=hip.EPI_GELU, aux=torch.empty_like(aux)), out)
    # and the quantised product is a faithful stand-in for the bf16 one (e4m3: 3 mantissa bits, errors average out over K)
    full = (a.float() @ b.float().t())
    assert rel(ops.gemm_nt_fp8(qa, sa, qb, sb), full) < 6e-2


def _site(gpu, scale):
    from mvuld_amd import ops
    st = ops.Fp8Site(torch.tensor([scale, 0.0], dtype=torch.float32, device=gpu))
    st.cal = True
    return st


@pytest.mark.parametrize("M,N,K", [(6401, 2056, 576), (25088, 3072, 768), (300, 264, 256)])
def test_gemm_nt_fp8_gelu_epilogue_emits_e4m3(gpu, M, N, K):
    """Fused quantisation in the fp8 GEMM's GELU epilogue: the e4m3 side output must be exactly the e4m3 conversion of the bf16
    activation the same launch stores (under the site's scale, saturating), max|activation| must land in the site's amax slot, and
    the fp8-only form (inference: no bf16 copy) must write the same bytes."""
    from mvuld_amd import ops, hip
    g = torch.Generator().manual_seed(17)
    a = ((torch.rand((M, K), generator=g) - 0.5) * 3).to(torch.bfloat16)
    b = ((torch.rand((N, K), generator=g) - 0.5) * 0.2).to(torch.bfloat16)
    bias = (torch.rand((N,), generator=g) - 0.5).to(gpu)
    qa, sa = ops.quant_fp8(a.to(gpu))
    qb, sb = ops.quant_fp8(b.to(gpu))
    plain = ops.gemm_nt_fp8(qa, sa, qb, sb, bias=bias, epi=hip.EPI_GELU)
    amax = float(plain.float().abs().max())
    for scale in (amax / 448.0, amax / 448.0 / 3.0):                     # the second one saturates a good part of the tensor
        site = _site(gpu, scale)
        aux = torch.empty((M, N), dtype=torch.bfloat16, device=gpu)
        out, (q, sc) = ops.gemm_nt_fp8(qa, sa, qb, sb, bias=bias, epi=hip.EPI_GELU, aux=aux, emit=site)
        assert torch.equal(out, plain) and sc.data_ptr() == site.state.data_ptr()
        want = (out.float() * (1.0 / torch.tensor(scale, dtype=torch.float32))).clamp(-448, 448).cpu().to(torch.float8_e4m3fn)
        got = q.cpu().view(torch.float8_e4m3fn)
        mism = float((got.float() != want.float()).float().mean())
        assert mism < 1e-4 and float((got.float() - want.float()).abs().max()) <= 32.0, mism
        assert float(site.state[1]) == amax and float(site.state[0]) == pytest.approx(scale)
        site.state[1] = 0.0
        none, (q2, _) = ops.gemm_nt_fp8(qa, sa, qb, sb, bias=bias, epi=hip.EPI_GELU, emit=site, need_out=False)
        assert none is None and torch.equal(q2, q) and float(site.state[1]) == amax


@pytest.mark.parametrize("rows,C,pre,res", [(6272, 256, False, True), (1000, 768, True, False), (333, 1024, False, True), (50, 512, False, False)])
def test_layernorm_fwd_q8_emits_e4m3(gpu, rows, C, pre, res):
    """LayerNorm with the fused e4m3 side output: y / mean / rstd identical to the plain launch, q = e4m3(y / scale) exactly, amax."""
    from mvuld_amd import ops
    g = torch.Generator().manual_seed(23)
    x = (torch.randn((rows, C), generator=g) * 2).to(torch.bfloat16).to(gpu)
    p = torch.randn((rows, C), generator=g).to(torch.bfloat16).to(gpu) if pre else None
    r = torch.randn((rows, C), generator=g).to(torch.bfloat16).to(gpu) if res else None
    gam, bet = (torch.rand(C, generator=g) + 0.5).to(gpu), (torch.rand(C, generator=g) - 0.5).to(gpu)
    y0, m0, r0, s0 = ops.layernorm_fwd(x, gam, bet, 1e-5, residual=r, pre=p, want_sum=pre)
    amax = float(y0.float().abs().max())
    site = _site(gpu, amax / 448.0 / 1.5)
    y, m, rs, s, (q, sc) = ops.layernorm_fwd(x, gam, bet, 1e-5, residual=r, pre=p, want_sum=pre, emit=site)
    assert torch.equal(y, y0) and torch.equal(m, m0) and torch.equal(rs, r0) and (not pre or torch.equal(s, s0))
    want = (y.float() * (1.0 / site.state[0:1])).clamp(-448, 448).cpu().to(torch.float8_e4m3fn)
    got = q.cpu().view(torch.float8_e4m3fn)
    assert float((got.float() != want.float()).float().mean()) < 1e-4 and float((got.float() - want.float()).abs().max()) <= 32.0
    assert float(site.state[1]) == amax
    # an uncalibrated site: plain launch + dynamic quantisation that leaves its scale in the slot
    fresh = ops.Fp8Site(torch.zeros(2, dtype=torch.float32, device=gpu))
    y2, _, _, _, (q2, sc2) = ops.layernorm_fwd(x, gam, bet, 1e-5, residual=r, pre=p, want_sum=pre, emit=fresh)
    assert fresh.cal and torch.equal(y2, y0) and float(fresh.state[0]) == pytest.approx(amax / 448.0, rel=1e-6) and float(fresh.state[1]) == 0.0
    assert float(q2.cpu().view(torch.float8_e4m3fn).float().abs().max()) == 448.0


def test_fp8_roll_scales(gpu):
    from mvuld_amd.hip import call, ptr
    st = torch.tensor([1.0, 0.0, 2.0, 896.0, 0.0, 4.48], dtype=torch.float32, device=gpu)
    call("fp8_roll_scales", ptr(st), 3)
    assert st.cpu().tolist() == pytest.approx([1.0, 0.0, 2.0, 0.0, 0.01, 0.0])


def test_quant_e4m3_batched_equals_per_tensor(gpu):
    """mvuld_quant_e4m3_batched (all fp8 weight copies after an optimizer step, three launches) against one mvuld_quant_e4m3 per
    tensor: same scales, same bytes -- tensors smaller than, equal to and larger than a block chunk, one all-zero."""
    from mvuld_amd import ops
    from mvuld_amd.hip import call, ptr
    g = torch.Generator().manual_seed(29)
    sizes = [8, 8192, 8200, 3 * 8192, 768 * 3072, 512 * 512]
    ws = [(torch.randn(n, generator=g) * (0.02 * (i + 1))).to(gpu) for i, n in enumerate(sizes)]
    ws[3].zero_()
    qs = [torch.empty(n, dtype=torch.uint8, device=gpu) for n in sizes]
    scs = [torch.empty(1, dtype=torch.float32, device=gpu) for _ in sizes]
    rows, b0 = [], 0
    for w, q, sc in zip(ws, qs, scs):
        rows.append([w.data_ptr(), q.data_ptr(), sc.data_ptr(), w.numel(), b0])
        b0 += (w.numel() + 8191) // 8192
    table = torch.tensor(rows, dtype=torch.int64).to(gpu)
    partials = torch.empty(b0, dtype=torch.float32, device=gpu)
    call("quant_e4m3_batched", ptr(table), len(sizes), b0, ptr(partials))
    for w, q, sc in zip(ws, qs, scs):
        q1, s1 = ops.quant_fp8(w)
        assert float(sc) == float(s1) and torch.equal(q, q1)


def test_gemm_nt_fp8_rejects_ineligible_shapes(gpu):
    from mvuld_amd import ops, hip
    qa = torch.zeros((300, 128), dtype=torch.uint8, device=gpu)
    qb = torch.zeros((64, 128), dtype=torch.uint8, device=gpu)
    s = torch.ones(1, device=gpu)
    with pytest.raises(RuntimeError, match="not eligible"):
        ops.gemm_nt_fp8(qa, s, qb, s)                      # K = 128 < 256
    assert not ops.fp8_eligible(300, 64, 128) and ops.fp8_eligible(300, 64, 256)


def test_fast_erf_gelu_epilogue_accuracy(gpu):
    """bf16-output GEMM epilogues use a 13-instruction erf (Abramowitz-Stegun 7.1.26, |err| <= 1.5e-7): GELU and dGELU through
    an identity product over a dense sweep of pre-activations must match torch's erf GELU to bf16 rounding."""
    from mvuld_amd import ops, hip
    n = 512
    xs = torch.linspace(-9.0, 9.0, n * n).view(n, n)
    X = xs.to(torch.bfloat16)
    eye = torch.eye(n, dtype=torch.bfloat16, device=gpu)
    for mode in (0, 2):
        hip.LIB.fn("mvuld_set_gemm_p256_mode")(mode)
        try:
            aux = torch.empty((n, n), dtype=torch.bfloat16, device=gpu)
            out = ops.gemm_nt(X.to(gpu), eye, epi=hip.EPI_GELU, aux=aux)
            ref = F.gelu(X.float())
            assert float((out.float().cpu() - ref).abs().max()) <= 2.0 ** -8 * 9.0 + 1e-6
            assert float(((out.float().cpu() - ref).abs() / (ref.abs() + 1e-3)).max()) < 1e-2
            p = X.float().clone().requires_grad_(True)
            F.gelu(p).sum().backward()
            ones = torch.ones((n, n), dtype=torch.bfloat16, device=gpu)
            out = ops.gemm_nt(ones, eye, epi=hip.EPI_MUL_DGELU, aux=X.to(gpu))     # (1 . I^T) * gelu'(x)
            assert float((out.float().cpu() - p.grad).abs().max()) < 1e-2
            act = ops.gemm_nt(X.to(gpu), eye, epi=hip.EPI_GELU_DG, aux=aux)          # gelu(x) with gelu'(x) as the side output
            assert float((act.float().cpu() - ref).abs().max()) <= 2.0 ** -8 * 9.0 + 1e-6
            assert float((aux.float().cpu() - p.grad).abs().max()) <= 2.0 ** -8 * 1.13 + 1e-6      # bf16 rounding of |gelu'| <= 1.13
        finally:
            hip.LIB.fn("mvuld_set_gemm_p256_mode")(1)


@pytest.mark.parametrize("dtype", DTYPES)
def test_transpose_colsum(gpu, dtype):
    from mvuld_amd import ops
    x = rt(T("tx", (3, 70, 130)), dtype)
    out = ops.transpose(dev(x, dtype), R=70, C=130, batch=3)
    assert torch.equal(out.float().cpu(), x.transpose(1, 2).contiguous())
    y = rt(T("cs", (1000, 96)), dtype)
    acc = torch.ones(32, device=gpu)
    ops.colsum_into(dev(y, dtype), acc, N=32, col0=64)
    assert rel(acc, y[:, 64:].sum(0) + 1) < 1e-4


# ------------------------------------------------------------------------------------------------ norms
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("C", [32, 128, 768, 1024])
def test_layernorm_fwd_bwd(gpu, dtype, C):
    from mvuld_amd import ops
    rows, rps = 48, 12
    x, res = rt(T("lx", (rows, C)), dtype), rt(T("lr", (rows, C)), dtype)
    gm, bt = torch.nn.Parameter(T("lg", (C,), 0.5, 1.5)), torch.nn.Parameter(T("lb", (C,)))
    rsc = T("lrs", (rows // rps,), 0.0, 2.0)
    dy = rt(T("ldy", (rows, C)), dtype)
    xr = x.clone().requires_grad_(True)
    ref = res + rsc.repeat_interleave(rps)[:, None] * F.layer_norm(xr, (C,), gm, bt, 1e-5)
    ref.backward(dy)
    G, Bt = torch.nn.Parameter(dev(gm.data)), torch.nn.Parameter(dev(bt.data))
    y, mean, rstd, _ = ops.layernorm_fwd(dev(x, dtype), G.data, Bt.data, 1e-5, residual=dev(res, dtype), rowscale=dev(rsc), rows_per_sample=rps)
    assert rel(y, ref) < tol(dtype)
    dx = ops.layernorm_bwd(dev(dy, dtype), dev(x, dtype), G, Bt, mean, rstd, dev(rsc), rps)
    assert rel(dx, xr.grad) < tol(dtype)
    assert rel(G.grad, gm.grad) < 1e-3 and rel(Bt.grad, bt.grad) < 1e-3
    # post-LN form: LN(x + pre)
    y2, _, _, xs = ops.layernorm_fwd(dev(x, dtype), G.data, Bt.data, 1e-5, pre=dev(res, dtype), want_sum=True)
    assert rel(y2, F.layer_norm(rt(x + res, dtype), (C,), gm, bt, 1e-5)) < tol(dtype)
    assert rel(xs, x + res) < tol(dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(8, 96), (6, 100, 20)])
@pytest.mark.parametrize("training", [True, False])
def test_batchnorm(gpu, dtype, shape, training):
    from mvuld_amd.models.GraphModel import _BatchNormFn
    C = shape[1]
    bn = torch.nn.BatchNorm1d(C)
    bn.weight.data = T("bw", (C,), 0.5, 1.5); bn.bias.data = T("bb", (C,))
    bn.running_mean.data = T("brm", (C,), -0.2, 0.2); bn.running_var.data = T("brv", (C,), 0.5, 1.5)
    bn.train(training)
    import copy
    gb = copy.deepcopy(bn).to(gpu)
    x = rt(T("bx", shape), dtype)
    dy = rt(T("bdy", shape), dtype)
    xr = x.clone().requires_grad_(True)
    ref = bn(xr)
    ref.backward(dy)
    xg = dev(x, dtype).requires_grad_(True)
    y = _BatchNormFn.apply(xg, gb, gb.weight)
    y.backward(dev(dy, dtype))
    assert rel(y, ref) < tol(dtype)
    assert rel(xg.grad, xr.grad) < tol(dtype)
    assert rel(gb.weight.grad, bn.weight.grad) < 2e-3 and rel(gb.bias.grad, bn.bias.grad) < 2e-3
    assert rel(gb.running_mean, bn.running_mean) < 1e-4 and rel(gb.running_var, bn.running_var) < 1e-4


# ------------------------------------------------------------------------------------------------ attention
def _swin_attn_ref(qkv, table16, ls, B, H, hd, res, ws, shift):
    from oracle import swin_ref
    C = H * hd
    x = qkv.view(B, res, res, 3 * C)
    if shift:
        x = torch.roll(x, (-shift, -shift), (1, 2))
    n = res // ws
    xw = x.view(B, n, ws, n, ws, 3 * C).permute(0, 1, 3, 2, 4, 5).reshape(B * n * n, ws * ws, 3, H, hd).permute(2, 0, 3, 1, 4)
    q, k, v = xw[0], xw[1], xw[2]
    att = F.normalize(q, dim=-1) @ F.normalize(k, dim=-1).transpose(-1, -2)
    att = att * torch.clamp(ls, max=math.log(100.0)).exp().view(1, H, 1, 1)
    N = ws * ws
    att = att + table16[swin_ref.rel_index(ws).reshape(-1)].view(N, N, H).permute(2, 0, 1)[None]
    if shift:
        m = swin_ref.shift_mask(res, ws, shift)
        att = (att.view(B, n * n, H, N, N) + m[None, :, None]).view(B * n * n, H, N, N)
    o = (att.softmax(-1) @ v).transpose(1, 2).reshape(B, n, n, ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(B, res, res, C)
    if shift:
        o = torch.roll(o, (shift, shift), (1, 2))
    return o.reshape(B * res * res, C)


@pytest.mark.parametrize("dtype,impl", [(torch.float32, "simple"), (torch.bfloat16, "simple"), (torch.bfloat16, "auto")])
@pytest.mark.parametrize("res,ws,shift,H", [(14, 7, 3, 2), (14, 7, 0, 2), (28, 14, 7, 1), (14, 14, 0, 4), (56, 28, 14, 2), (28, 28, 0, 3)])
def test_swin_window_attention(gpu, dtype, impl, res, ws, shift, H):
    """impl "auto" = the MFMA kernels (bf16), "simple" = the VALU kernels."""
    from mvuld_amd import ops
    B, hd = 2, 32
    C = H * hd
    qkv = rt(T("aq", (B * res * res, 3 * C), -2, 2), dtype)
    T2 = (2 * ws - 1) ** 2
    table = T("at", (T2, H), 0.0, 16.0)
    ls = T("als", (H,), 1.5, 5.0)          # some heads above the ln(100) clamp
    dout = rt(T("ado", (B * res * res, C)), dtype)
    q_, t_, l_ = qkv.clone().requires_grad_(True), table.clone().requires_grad_(True), ls.clone().requires_grad_(True)
    ref = _swin_attn_ref(q_, t_, l_, B, H, hd, res, ws, shift)
    ref.backward(dout)
    g = ops.AttnGeom(0, B, H, hd, ws * ws, (res // ws) ** 2, res, ws, shift)
    ops.ATTN_IMPL[0] = impl
    try:
        gq, gt, gl, gdo = dev(qkv, dtype), dev(table), dev(ls), dev(dout, dtype)
        out, lse = ops.attn_fwd(g, gq, gt, gl)
        # MFMA path: q~ = tau*q^ and k^ are re-rounded to bf16 for the matrix cores (as autocast would); with this test's
        # logit scales up to the clamp (tau = 100) a 2^-9 operand error is a ~0.05 score error, hence the wider bound
        assert rel(out, ref) < (tol(dtype) if impl == "simple" else 6e-2)
        dtab = torch.zeros((T2, H), device=gpu)
        dls = torch.zeros(H, device=gpu)
        dqkv = ops.attn_bwd(g, gq, out, gdo, lse, gt, gl, None, dtab, dls)
    finally:
        ops.ATTN_IMPL[0] = "auto"
    k = 2 if impl == "simple" else 4
    assert rel(dqkv, q_.grad) < tol(dtype) * k
    assert rel(dtab, t_.grad) < tol(dtype) * k
    # d(logit_scale) is a sum of ~1e6 cancelling terms; with tau near the clamp the bf16 operand rounding of the matrix-core
    # path moves it by 10-20 % (at the reference's initial tau = 10 the two implementations agree to <1 %: tools/diag_attn.py)
    assert rel(dls, l_.grad) < (tol(dtype) * k if impl == "simple" else 0.3)


@pytest.mark.parametrize("res,ws,shift,H", [(56, 28, 14, 2), (28, 28, 0, 3), (16, 8, 4, 2), (24, 12, 6, 1), (8, 4, 0, 2), (40, 20, 10, 1), (112, 28, 14, 1)])
def test_window_fast_path_forward(gpu, res, ws, shift, H):
    """Round 3: the forward window fast path (attention_mfma.hip attn_fwd_win_k: three shifted copies of the bias table read with aligned
    8-byte LDS reads, one offset word read per block, two query tiles per wave) against the general kernel (mvuld_set_attn_win(0)).
    Mode 2 keeps the general kernel's arithmetic, block sizes and accumulation order: out and lse must be equal BIT FOR BIT.  Mode 1
    (default) defers the running maximum (it moves only when a block exceeds it by 2^6), which changes the scale at which P is rounded to
    bf16: out agrees to bf16 rounding, lse to fp32 rounding.  Geometries: shifted (masked last row / column of windows) and unshifted,
    N a multiple of 64 (no tail block), tails of 16 + 16 padding positions, a window smaller than one block, an odd tile count (a duplicate
    slot in the last pair) and stage 0's 16 windows."""
    from mvuld_amd import ops, hip
    B, hd = 2, 32
    C = H * hd
    T2 = (2 * ws - 1) ** 2
    qkv = dev(rt(T("wq", (B * res * res, 3 * C), -2, 2), torch.bfloat16), torch.bfloat16)
    table, ls = dev(T("wt", (T2, H), 0.0, 16.0)), dev(T("wls", (H,), 1.5, 5.0))
    g = ops.AttnGeom(0, B, H, hd, ws * ws, (res // ws) ** 2, res, ws, shift)
    got = {}
    try:
        for mode in (0, 1, 2):
            hip.LIB.fn("mvuld_set_attn_win")(mode)
            out, lse = ops.attn_fwd(g, qkv, table, ls)
            torch.cuda.synchronize()
            got[mode] = (out, lse)
    finally:
        hip.LIB.fn("mvuld_set_attn_win")(1)
    assert torch.equal(got[2][0], got[0][0]) and torch.equal(got[2][1], got[0][1]), (rel(got[2][0], got[0][0]), rel(got[2][1], got[0][1]))
    assert rel(got[1][0], got[0][0]) < 1e-2 and rel(got[1][1], got[0][1]) < 1e-4, (rel(got[1][0], got[0][0]), rel(got[1][1], got[0][1]))
    assert bool(torch.isfinite(got[1][0].float()).all()) and bool(torch.isfinite(got[1][1]).all())


@pytest.mark.parametrize("res,ws,shift,H,ls_hi", [(56, 28, 14, 2, 3.0), (112, 28, 14, 1, 3.0), (16, 8, 4, 2, 3.0), (24, 12, 6, 1, 3.0), (28, 14, 7, 2, 3.0),
                                                  (56, 28, 14, 2, 5.0), (28, 28, 0, 2, 3.0)])
def test_shifted_window_vertical_mask_split_skip_is_bit_identical(gpu, res, ws, shift, H, ls_hi):
    """In a shifted block the windows of the last row hold two vertical mask regions; every pair across the split carries the -100 of
    swin_transformer_v2.py:245-268.  The forward, dQ and dK/dV passes skip the tiles made of such pairs only (am_ysplit, heads
    with tau <= 22: the skipped probabilities are < 2^-57 of their row's largest); mvuld_set_attn_yskip(0) computes every pair.  All
    outputs must be EQUAL bit for bit (out, lse, dqkv) or to the rounding of their atomic sums (d(bias table), d(logit_scale)).  ls_hi = 5.0
    draws heads past the tau bound (which must then compute every pair: equality again), shift 0 has no split."""
    from mvuld_amd import ops, hip
    B, hd = 2, 32
    C = H * hd
    T2 = (2 * ws - 1) ** 2
    qkv = dev(rt(T("yq", (B * res * res, 3 * C), -2, 2), torch.bfloat16), torch.bfloat16)
    dout = dev(rt(T("yd", (B * res * res, C)), torch.bfloat16), torch.bfloat16)
    table, ls = dev(T("yt", (T2, H), 0.0, 16.0)), dev(T("yls", (H,), 1.5, ls_hi))
    g = ops.AttnGeom(0, B, H, hd, ws * ws, (res // ws) ** 2, res, ws, shift)
    got = {}
    try:
        for on in (1, 0):
            hip.LIB.fn("mvuld_set_attn_yskip")(on)
            out, lse = ops.attn_fwd(g, qkv, table, ls)
            dtab, dls = torch.zeros((T2, H), device=gpu), torch.zeros(H, device=gpu)
            dqkv = ops.attn_bwd(g, qkv, out, dout, lse, table, ls, None, dtab, dls)
            torch.cuda.synchronize()
            got[on] = (out, lse, dqkv, dtab, dls)
    finally:
        hip.LIB.fn("mvuld_set_attn_yskip")(1)
    names = ("out", "lse", "dqkv")
    for n, a, b in zip(names, got[1][:3], got[0][:3]):
        assert torch.equal(a, b), (n, rel(a, b))
    # d(bias table) and d(logit_scale) are sums of float atomics (LDS / global), whose order is not fixed from run to run
    assert rel(got[1][3], got[0][3]) < 1e-6 and rel(got[1][4], got[0][4]) < 1e-6
    assert bool(torch.isfinite(got[1][2].float()).all())


@pytest.mark.parametrize("res,ws,shift,H", [(56, 28, 14, 2), (28, 28, 0, 3), (28, 14, 7, 2), (16, 8, 4, 2)])
def test_window_attention_skips_droppath_dropped_samples(gpu, res, ws, shift, H):
    """sample_scale = the block's per-sample DropPath factors (swin_transformer_v2.py:301).  Samples whose factor is 0 are not computed:
    their `out` / lse rows and dqkv rows are zeros, they add nothing to d(bias table) / d(logit_scale); every other sample's rows are
    bit-identical to a call without the vector.  (Downstream the dropped samples' branch is multiplied by 0 and their d(out) IS 0, which
    is what the reference computes the long way; here d(out) of the dropped samples is zeroed for the comparison run.)"""
    from mvuld_amd import ops
    B, hd = 5, 32
    C = H * hd
    L = res * res
    T2 = (2 * ws - 1) ** 2
    qkv = dev(rt(T("sq", (B * L, 3 * C), -2, 2), torch.bfloat16), torch.bfloat16)
    dout = dev(rt(T("sd", (B * L, C)), torch.bfloat16), torch.bfloat16)
    table, ls = dev(T("st", (T2, H), 0.0, 16.0)), dev(T("sls", (H,), 1.5, 3.0))
    scale = torch.tensor([1.25, 0.0, 1.25, 0.0, 1.25], device=gpu)
    keep = (scale != 0).repeat_interleave(L)
    g = ops.AttnGeom(0, B, H, hd, ws * ws, (res // ws) ** 2, res, ws, shift)

    def run(ss, do):
        out, lse = ops.attn_fwd(g, qkv, table, ls, sample_scale=ss)
        dtab, dls = torch.zeros((T2, H), device=gpu), torch.zeros(H, device=gpu)
        dqkv = ops.attn_bwd(g, qkv, out, do, lse, table, ls, None, dtab, dls, sample_scale=ss)
        torch.cuda.synchronize()
        return out, lse.view(B, -1), dqkv, dtab, dls
    assert ops.USE_DROPPATH_SKIP[0]
    full = run(None, dout * keep.view(-1, 1).to(dout.dtype))
    skip = run(scale, dout)                                  # d(out) of the dropped samples is never read
    assert torch.equal(skip[0][keep], full[0][keep]) and float(skip[0][~keep].float().abs().max()) == 0.0
    kb = scale != 0
    assert torch.equal(skip[1][kb], full[1][kb]) and float(skip[1][~kb].abs().max()) == 0.0
    assert torch.equal(skip[2][keep], full[2][keep]) and float(skip[2][~keep].float().abs().max()) == 0.0
    assert rel(skip[3], full[3]) < 1e-5 and rel(skip[4], full[4]) < 1e-5          # sums of float atomics: order not fixed


@pytest.mark.parametrize("res,ws,shift,H,ls_lo,ls_hi", [(56, 28, 14, 2, 1.5, 3.0), (28, 28, 0, 3, 1.5, 5.0), (112, 28, 14, 1, 1.5, 3.0), (16, 8, 4, 2, 1.5, 3.0),
                                                        (24, 12, 6, 1, 1.5, 3.0), (8, 4, 0, 2, 1.5, 3.0), (40, 20, 10, 1, 1.5, 5.0), (48, 24, 12, 2, 2.0, 2.5)])
def test_fused_window_backward(gpu, res, ws, shift, H, ls_lo, ls_hi):
    """Round 4: attn_bwd_fused_win_k -- dQ, dK, dV, d(bias table) and d(logit_scale) from ONE recomputation of the scores per (query, key)
    pair (row-aligned 32 x 32 blocks, key rows resident per wave, dS through LDS once for dQ, the table gradient folded with whole-wave DPP
    shifts) -- against (a) the fp32 restatement of swin_transformer_v2.py:140-179,245-268 at the bounds of test_swin_window_attention and
    (b) the three-pass matrix-core backward (mvuld_set_attn_bwd_fused(0)), which rounds the same operands to bf16: the two agree to the
    rounding of bf16 P / dS and of the order of the fp32 sums.  Geometries: the three SwinV2-base window stages' shapes (one window, 4 and 16
    windows, shifted and not), windows with fewer rows than the waves own (4, 8, 12, 20, 24: waves with 0-6 key rows), heads past the
    ln(100) clamp (no d(logit_scale) there) and past the tau bound of the vertical-split skip."""
    from mvuld_amd import ops, hip
    B, hd = 2, 32
    C = H * hd
    T2 = (2 * ws - 1) ** 2
    qkv = rt(T("fq", (B * res * res, 3 * C), -2, 2), torch.bfloat16)
    table, ls = T("ft", (T2, H), 0.0, 16.0), T("fls", (H,), ls_lo, ls_hi)
    dout = rt(T("fdo", (B * res * res, C)), torch.bfloat16)
    q_, t_, l_ = qkv.clone().requires_grad_(True), table.clone().requires_grad_(True), ls.clone().requires_grad_(True)
    ref = _swin_attn_ref(q_, t_, l_, B, H, hd, res, ws, shift)
    ref.backward(dout)
    g = ops.AttnGeom(0, B, H, hd, ws * ws, (res // ws) ** 2, res, ws, shift)
    gq, gt, gl, gdo = dev(qkv, torch.bfloat16), dev(table), dev(ls), dev(dout, torch.bfloat16)
    out, lse = ops.attn_fwd(g, gq, gt, gl)
    assert hip.LIB.fn("mvuld_attn_bwd_fused_active")(0, hd, ws) == 1
    got = {}
    try:
        for fused in (1, 0):
            hip.LIB.fn("mvuld_set_attn_bwd_fused")(fused)
            dtab, dls = torch.zeros((T2, H), device=gpu), torch.zeros(H, device=gpu)
            dqkv = ops.attn_bwd(g, gq, out, gdo, lse, gt, gl, None, dtab, dls)
            torch.cuda.synchronize()
            got[fused] = (dqkv, dtab, dls)
    finally:
        hip.LIB.fn("mvuld_set_attn_bwd_fused")(1)
    k = 4
    for part, (lo, hi) in {"dq": (0, C), "dk": (C, 2 * C), "dv": (2 * C, 3 * C)}.items():
        assert bool(torch.isfinite(got[1][0].float()).all())
        assert rel(got[1][0][:, lo:hi], q_.grad[:, lo:hi]) < tol(torch.bfloat16) * k, (part, rel(got[1][0][:, lo:hi], q_.grad[:, lo:hi]))
        assert rel(got[1][0][:, lo:hi], got[0][0][:, lo:hi]) < 2e-2, (part, rel(got[1][0][:, lo:hi], got[0][0][:, lo:hi]))
    assert rel(got[1][1], t_.grad) < tol(torch.bfloat16) * k, rel(got[1][1], t_.grad)
    assert rel(got[1][1], got[0][1]) < 1e-2, rel(got[1][1], got[0][1])
    assert rel(got[1][2], l_.grad) < 0.3 and rel(got[1][2], got[0][2]) < 5e-2, (rel(got[1][2], l_.grad), rel(got[1][2], got[0][2]))


@pytest.mark.parametrize("dtype,impl", [(torch.float32, "simple"), (torch.bfloat16, "simple"), (torch.bfloat16, "auto")])
@pytest.mark.parametrize("hd,L", [(32, 100), (64, 100), (64, 512)])
def test_padmask_attention(gpu, dtype, impl, hd, L):
    from mvuld_amd import ops
    B, H = 3, 2
    C = H * hd
    lens = [L, 37, 5]
    valid = torch.zeros(B, L, dtype=torch.int32)
    for i, n in enumerate(lens):
        valid[i, :n] = 1
    qkv = rt(T("pq", (B * L, 3 * C), -2, 2), dtype)
    dout = rt(T("pdo", (B * L, C)), dtype) * valid.view(-1, 1)
    q_ = qkv.clone().requires_grad_(True)
    x = q_.view(B, L, 3, H, hd).permute(2, 0, 3, 1, 4)
    m = valid.bool()
    add = (1.0 - (m[:, None, :] & m[:, :, None]).float()[:, None]) * -10000.0
    s = x[0] @ x[1].transpose(-1, -2) / math.sqrt(hd) + add
    ref = (s.softmax(-1) @ x[2]).transpose(1, 2).reshape(B * L, C)
    ref.backward(dout)
    g = ops.AttnGeom(1, B, H, hd, L, 1, 0, 0, 0, 1.0 / math.sqrt(hd))
    vm = valid.view(-1, 1).float()
    ops.ATTN_IMPL[0] = impl
    try:
        gq, gv, gdo = dev(qkv, dtype), dev(valid), dev(dout, dtype)
        out, lse = ops.attn_fwd(g, gq, valid=gv)
        assert rel(out.float().cpu() * vm, ref * vm) < tol(dtype)
        dqkv = ops.attn_bwd(g, gq, out, gdo, lse, valid=gv)
    finally:
        ops.ATTN_IMPL[0] = "auto"
    assert rel(dqkv.float().cpu() * vm, q_.grad * vm) < tol(dtype) * 2



def test_cpb_table(gpu):
    from mvuld_amd import hip
    from mvuld_amd.hip import call, ptr
    from oracle import swin_ref
    ws, H = 7, 4
    coords = swin_ref.coords_table(ws, 6)
    T2 = coords.shape[0]
    W1 = T("c1", (512, 2)).requires_grad_(True); b1 = T("cb", (512,)).requires_grad_(True); W2 = T("c2", (H, 512), -0.2, 0.2).requires_grad_(True)
    ref = 16 * torch.sigmoid(F.linear(F.relu(F.linear(coords, W1, b1)), W2))
    dt_ = T("cdt", (T2, H))
    ref.backward(dt_)
    hid = torch.empty((T2, 512), device=gpu); tab = torch.empty((T2, H), device=gpu)
    # keep every device operand alive in a named variable (a temporary's block can be re-used by the next upload)
    gc, gW1, gb1, gW2, gdt = dev(coords), dev(W1.data), dev(b1.data), dev(W2.data), dev(dt_)
    call("cpb_table_fwd", ptr(gc), ptr(gW1), ptr(gb1), ptr(gW2), ptr(hid), ptr(tab), T2, H)
    assert rel(tab, ref) < 1e-5
    scratch = torch.empty(((T2 + 15) // 16) * (32 * 512 + 1536), device=gpu)
    for ws_ in (None, scratch):             # atomics path, then per-block partials + reduce kernel
        d1, db, d2 = torch.ones(512, 2, device=gpu), torch.ones(512, device=gpu), torch.ones(H, 512, device=gpu)
        call("cpb_table_bwd", ptr(gc), ptr(gW2), ptr(hid), ptr(tab), ptr(gdt), ptr(d1), ptr(db), ptr(d2), T2, H,
             ptr(ws_), ws_.numel() * 4 if ws_ is not None else 0)
        assert rel(d1, W1.grad + 1) < 1e-4 and rel(db, b1.grad + 1) < 1e-4 and rel(d2, W2.grad + 1) < 1e-4


# ------------------------------------------------------------------------------------------------ graph
@pytest.mark.parametrize("dtype", DTYPES)
def test_gatconv(gpu, dtype):
    from mvuld_amd.models.GraphModel import GATConv
    from mvuld_amd.graph import BatchedGraph
    from oracle import head_ref
    N, Fin, H, O = 37, 64, 4, 96
    # multigraph: duplicate edges, duplicate self loops, in-degree-1 nodes
    src = torch.tensor([0, 0, 1, 2, 2, 5, 5, 5, 9, 9] + list(range(N)) + [3, 3], dtype=torch.int64)
    dst = torch.tensor([1, 1, 2, 3, 3, 6, 6, 7, 9, 9] + list(range(N)) + [3, 3], dtype=torch.int64)
    g = BatchedGraph(src, dst, [20, 17]).to(gpu)
    conv = GATConv(Fin, O, H, feat_drop=0.0)
    sd = {"fc.weight": T("gw", (H * O, Fin), -0.2, 0.2), "attn_l": T("gl", (1, H, O), -0.3, 0.3),
          "attn_r": T("gr", (1, H, O), -0.3, 0.3), "bias": T("gbb", (H * O,))}
    conv.load_state_dict(sd)
    conv = conv.to(gpu)
    x = rt(T("gx", (N, Fin)), dtype)
    dy = rt(T("gdy", (N, H, O)), dtype)
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    if dtype == torch.bfloat16:
        sdr["fc.weight"] = rt(sd["fc.weight"], dtype).requires_grad_(True)
    xr = x.clone().requires_grad_(True)
    ref = head_ref.gat_conv(sdr, "", xr, src, dst, H, O)
    ref.backward(dy)
    xg = dev(x, dtype).requires_grad_(True)
    out = conv(g, xg)
    out.backward(dev(dy, dtype))
    assert rel(out, ref) < tol(dtype)
    assert rel(xg.grad, xr.grad) < tol(dtype) * 2
    for k, p in (("fc.weight", conv.fc.weight), ("attn_l", conv.attn_l), ("attn_r", conv.attn_r), ("bias", conv.bias)):
        assert rel(p.grad, sdr[k].grad) < tol(dtype) * 2, k


@pytest.mark.parametrize("dtype", DTYPES)
def test_segment_pad_l2norm_meanpool(gpu, dtype):
    from mvuld_amd.models.GraphModel import _SegmentPadFn, _L2NormMeanFn
    from mvuld_amd.models.unixcoder import _MaskedMeanFn
    from oracle import head_ref
    bnn = [60, 100, 130, 7]
    off = torch.tensor([0] + list(torch.tensor(bnn).cumsum(0)), dtype=torch.int32)
    h = rt(T("sp", (sum(bnn), 24)), dtype)
    hr = h.clone().requires_grad_(True)
    ref = head_ref.unbatch_pad(hr, bnn)
    dy = rt(T("spd", tuple(ref.shape)), dtype)
    ref.backward(dy)
    hg = dev(h, dtype).requires_grad_(True)
    out = _SegmentPadFn.apply(hg, dev(off), len(bnn), 100)
    out.backward(dev(dy, dtype))
    assert torch.equal(out.float().cpu(), ref.detach()) and torch.equal(hg.grad.float().cpu(), hr.grad)
    B, Nn, C = 3, 100, 40
    g_ = rt(T("l2", (B, Nn, C)), dtype)
    gr = g_.clone().requires_grad_(True)
    r = (gr / gr.pow(2).sum(1, keepdim=True).sqrt()).mean(1)
    d = rt(T("l2d", (B, C)), dtype)
    r.backward(d)
    gg = dev(g_, dtype).view(B * Nn, C).requires_grad_(True)
    o = _L2NormMeanFn.apply(gg, B)
    o.backward(dev(d, dtype))
    assert rel(o, r) < tol(dtype) and rel(gg.grad.view(B, Nn, C), gr.grad) < tol(dtype)
    L = 50
    valid = torch.zeros(B, L, dtype=torch.int32); valid[0, :50] = 1; valid[1, :13] = 1; valid[2, :1] = 1
    t_ = rt(T("mp", (B, L, C)), dtype)
    tr = t_.clone().requires_grad_(True)
    m = valid.float()
    r = (tr * m[..., None]).sum(1) / m.sum(-1)[..., None]
    r.backward(d)
    tg = dev(t_, dtype).view(B * L, C).requires_grad_(True)
    o = _MaskedMeanFn.apply(tg, dev(valid), B, L)
    o.backward(dev(d, dtype))
    assert rel(o, r) < tol(dtype) and rel(tg.grad.view(B, L, C), tr.grad) < tol(dtype)


def test_cross_entropy_adamw_sumsq(gpu):
    from mvuld_amd.models.GraphModel import cross_entropy
    from mvuld_amd.hip import call, ptr
    lg = T("ce", (9, 2), -3, 3)
    tg = synth.ints("kt/cet", (9,), 0, 2)
    lr_ = lg.clone().requires_grad_(True)
    ref = F.cross_entropy(lr_, tg)
    ref.backward()
    lgpu = dev(lg).requires_grad_(True)
    loss, probs = cross_entropy(lgpu, dev(tg))
    loss.backward()
    assert rel(loss, ref) < 1e-5 and rel(probs, lg.softmax(1)) < 1e-5 and rel(lgpu.grad, lr_.grad) < 1e-5
    n = 10007
    p0, g0 = T("ap", (n,)), T("ag", (n,))
    p = torch.nn.Parameter(p0.clone()); p.grad = g0.clone() * 0.5
    opt = torch.optim.AdamW([p], lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.005)
    P, G = dev(p0.clone()), dev(g0.clone())
    M_, V_ = torch.zeros(n, device=gpu), torch.zeros(n, device=gpu)
    P16 = torch.empty(n, dtype=torch.bfloat16, device=gpu)
    ss = torch.zeros(1, device=gpu); no = torch.zeros(2, device=gpu)
    part = torch.empty(2048, device=gpu)
    call("sumsq", ptr(G), n, ptr(part), ptr(ss))
    big = torch.randn(3_000_001, device=gpu)                   # bit-reproducible: data-parallel ranks must agree on the clip coefficient
    r = [torch.zeros(1, device=gpu) for _ in range(3)]
    for ri in r:
        call("sumsq", ptr(big), big.numel(), ptr(part), ptr(ri))
    assert float(r[0]) == float(r[1]) == float(r[2]) and rel(r[0], (big.double() ** 2).sum().float()) < 1e-5
    assert rel(ss, (g0 ** 2).sum()) < 1e-5
    call("clip_coef", ptr(ss), float(g0.norm() * 0.5), 1.0, ptr(no))
    for step in (1, 2, 3):
        opt.step()
        call("adamw", ptr(P), ptr(G), ptr(M_), ptr(V_), ptr(P16), n, 1e-2, 0.9, 0.999, 1e-8, 0.005, step, ptr(no), 0, None)
    assert rel(no[0], g0.norm()) < 1e-5 and abs(float(no[1]) - 0.5) < 1e-4
    assert rel(P, p.data) < 1e-5 and rel(P16, p.data) < 1e-2
    assert float(G.abs().max()) > 0                     # zero_grad = 0 leaves the gradient alone ...
    call("adamw", ptr(P), ptr(G), ptr(M_), ptr(V_), ptr(P16), n, 1e-2, 0.9, 0.999, 1e-8, 0.005, 4, ptr(no), 1, None)
    assert float(G.abs().max()) == 0.0                  # ... zero_grad = 1 clears it in the same pass


@pytest.mark.parametrize("dtype", DTYPES)
def test_embed_im2col_patchmerge_dropout(gpu, dtype):
    from mvuld_amd.hip import call, ptr, dt
    B, L, Hd, V, MP = 2, 12, 40, 50, 20
    ids = synth.ints("kt/ids", (B, L), 2, V); ids[0, 8:] = 1; ids[1, 3:] = 1
    word, posw, typ = T("ew", (V, Hd)), T("ep", (MP, Hd)), T("et", (3, Hd))
    pos = torch.empty((B, L), dtype=torch.int32, device=gpu); valid = torch.empty_like(pos)
    gids, gword, gposw, gtyp = dev(ids), dev(word), dev(posw), dev(typ)
    call("position_ids", ptr(gids), ptr(pos), ptr(valid), B, L, 1)
    m = ids.ne(1).long()
    assert torch.equal(pos.cpu().long(), torch.cumsum(m, 1) * m + 1) and torch.equal(valid.cpu().long(), m)
    out = torch.empty((B * L, Hd), dtype=dtype, device=gpu)
    call("embed_fwd", ptr(gids), ptr(pos), ptr(gword), ptr(gposw), ptr(gtyp), ptr(out), B * L, Hd, V, MP, dt(out))
    ref = word[ids] + posw[pos.cpu().long()] + typ[0]
    assert rel(out.view(B, L, Hd), ref) < tol(dtype)
    img = T("im", (2, 3, 16, 16))
    cols = torch.empty((2 * 16, 48), dtype=dtype, device=gpu)
    gimg = dev(img)
    call("im2col_patch4", ptr(gimg), ptr(cols), 2, 16, dt(cols))
    w = T("imw", (8, 3, 4, 4))
    ref = F.conv2d(rt(img, dtype), w, stride=4).flatten(2).transpose(1, 2).reshape(32, 8)
    assert rel(cols.float().cpu() @ w.view(8, 48).t(), ref) < 1e-4
    x = rt(T("pm", (2, 6, 6, 5)), dtype)
    y = torch.empty((2 * 9, 20), dtype=dtype, device=gpu)
    gx = dev(x, dtype)
    call("patch_merge_gather", ptr(gx), ptr(y), 2, 6, 5, 0, dt(y))
    ref = torch.cat([x[:, 0::2, 0::2], x[:, 1::2, 0::2], x[:, 0::2, 1::2], x[:, 1::2, 1::2]], -1).reshape(18, 20)
    assert torch.equal(y.float().cpu(), ref)
    back = torch.empty((2 * 36, 5), dtype=dtype, device=gpu)
    call("patch_merge_gather", ptr(y), ptr(back), 2, 6, 5, 1, dt(y))
    assert torch.equal(back.float().cpu().view(2, 6, 6, 5), x)
    from mvuld_amd import ops
    big = torch.ones(200000, dtype=dtype, device=gpu)
    d1, d2 = ops.dropout(big, 0.2, 1234), ops.dropout(big, 0.2, 1234)
    assert torch.equal(d1, d2) and abs(float((d1 == 0).float().mean()) - 0.2) < 0.01
    assert abs(float(d1.float().mean()) - 1.0) < 0.02


@pytest.mark.parametrize("mode", ["tn256", "slab128", "atomic"])
@pytest.mark.parametrize("M,N,K", [(4096, 128, 128), (1000, 384, 136), (777, 72, 200), (20000, 256, 512), (16384, 776, 392),
                                   (16384, 768, 768), (6272, 1000, 760), (25088, 512, 2048), (2048, 256, 256), (9917, 768, 3072), (1001, 1024, 520),
                                   (70000, 1024, 256), (66001, 768, 256)])
def test_gemm_tn_wgrad(gpu, M, N, K, mode):
    """dW += dY^T X and db += colsum(dY) through the transpose-free matrix-core kernels (bf16 operands).  Three ways of combining the
    split token contraction: "tn256" = the 256 x 256-tile LDS-DMA kernel with slab partials + reduction launch where the shape
    fills its tiles (M % 32 == 0; the others fall through), "slab128" = 128 x 128 tiles with the last-arriver slab reduction
    (tickets must be back at zero for the next launch on the stream: three launches accumulate three times), "atomic" = fp32
    atomics from every split."""
    from mvuld_amd import ops, hip
    dy, x = rt(T("tn_dy", (M, N)), torch.bfloat16), rt(T("tn_x", (M, K)), torch.bfloat16)
    w = torch.nn.Parameter(torch.zeros(N, K, device=gpu))
    b = torch.nn.Parameter(torch.zeros(N, device=gpu))
    w.grad = torch.ones(N, K, device=gpu)
    b.grad = torch.ones(N, device=gpu)
    gdy, gx = dev(dy, torch.bfloat16), dev(x, torch.bfloat16)
    ops.USE_TN_SLABS[0] = mode != "atomic"
    hip.LIB.fn("mvuld_set_gemm_tn256")(1 if mode == "tn256" else 0)
    try:
        ops.linear_wgrad(gdy, gx, w, b)
        ref = dy.t() @ x
        assert rel(w.grad, ref + 1.0) < 2e-3
        assert rel(b.grad, dy.sum(0) + 1.0) < 2e-3
        ops.linear_wgrad(gdy, gx, w, b)
        ops.linear_wgrad(gdy, gx, w, b)
        assert rel(w.grad, 3.0 * ref + 1.0) < 2e-3
        assert rel(b.grad, 3.0 * dy.sum(0) + 1.0) < 2e-3
        tn, tk = -(-N // 256), -(-K // 256)
        enough = tn * tk >= 8 or (tn * tk >= 3 and M >= 65536)       # (3- and 4-tile weights only with a very long contraction)
        if mode == "tn256" and M >= 256 and enough and tn * tk * 65536 * 4 <= N * K * 5:      # tn256_plan(): the kernel takes these
            # ping-pong (default) and lockstep schedules of the 256 x 256 kernel contract in the same order: equal weight gradients, and
            # equal again on a repeated launch (a mis-placed wait reads a slab before its DMA has landed); the bias gradient is fp32
            # atomics from the splits, equal up to their order
            res = []
            for on in (1, 0, 1):
                hip.LIB.fn("mvuld_set_gemm_tn256_pingpong")(on)
                w.grad.zero_()
                b.grad.zero_()
                ops.linear_wgrad(gdy, gx, w, b)
                res.append((w.grad.clone(), b.grad.clone()))
            assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][0], res[2][0])
            assert rel(res[0][1], res[1][1]) < 1e-5
    finally:
        ops.USE_TN_SLABS[0] = True
        hip.LIB.fn("mvuld_set_gemm_tn256")(1)
        hip.LIB.fn("mvuld_set_gemm_tn256_pingpong")(1)


@pytest.mark.parametrize("rows,C", [(1000, 768), (333, 128), (4096, 1024)])
def test_layernorm_with_fused_hidden_dropout_is_bit_identical(gpu, rows, C):
    """mvuld_layernorm_fwd_drop = LayerNorm(dropout(x) + pre) in one pass (RobertaSelfOutput / RobertaOutput: dense -> dropout -> LayerNorm(+
    input)) must reproduce mvuld_dropout followed by mvuld_layernorm_fwd bit for bit: same counter hash, same bf16 rounding points."""
    from mvuld_amd import ops
    g_ = torch.Generator().manual_seed(rows + C)
    x = torch.randn(rows, C, generator=g_).to(torch.bfloat16).to(gpu)
    pre = torch.randn(rows, C, generator=g_).to(torch.bfloat16).to(gpu)
    gamma, beta = (torch.rand(C, generator=g_) + 0.5).to(gpu), torch.randn(C, generator=g_).to(gpu)
    for p, seed in ((0.1, 12345), (0.5, 7)):
        ops.USE_LN_DROP[0] = True
        y1, m1, r1, s1 = ops.layernorm_dropout_fwd(x, gamma, beta, 1e-5, pre, p, seed)
        ops.USE_LN_DROP[0] = False
        try:
            y0, m0, r0, s0 = ops.layernorm_dropout_fwd(x, gamma, beta, 1e-5, pre, p, seed)
        finally:
            ops.USE_LN_DROP[0] = True
        assert torch.equal(y1, y0) and torch.equal(s1, s0) and torch.equal(m1, m0) and torch.equal(r1, r0)
        assert not torch.equal(s1, (x.float() + pre.float()).to(torch.bfloat16))          # the mask did something
        # ... and the backward: mvuld_layernorm_bwd_drop = layernorm_bwd + dropout of its output, one pass, the same bits (parameter
        # gradients included: the column sums do not see the mask)
        gp = [torch.nn.Parameter(gamma.clone()), torch.nn.Parameter(beta.clone())]
        outs = []
        for fused in (True, False):
            ops.USE_LN_DROP[0] = fused
            try:
                for q in gp:
                    q.grad = torch.zeros_like(q)
                dx, dxd = ops.layernorm_bwd_dropout(pre, s1, gp[0], gp[1], m1, r1, p, seed + 1)
                outs.append((dx, dxd, gp[0].grad.clone(), gp[1].grad.clone()))
            finally:
                ops.USE_LN_DROP[0] = True
        assert all(torch.equal(a, b) for a, b in zip(outs[0][:2], outs[1][:2]))
        assert all(rel(a, b) < 1e-6 for a, b in zip(outs[0][2:], outs[1][2:]))
        assert float((outs[0][1] == 0).float().mean()) > 0.5 * p and not torch.equal(outs[0][0], outs[0][1])


@pytest.mark.parametrize("rows,C", [(6272, 512), (1568, 1024), (25088, 128), (333, 256)])
def test_layernorm_backward_with_deferred_parameter_gradients(gpu, rows, C):
    """While a weight-gradient stream is active, layernorm_bwd writes dx and its column partials only; the reductions into d gamma / d beta
    are issued together, in one launch on that stream, when the encoder's backward-done hook fires or the stream is joined
    (mvuld_layernorm_bwd_nparts / _reduce_batch).  Same dx bits, same gradients, nothing added before the flush, several LayerNorms per
    launch, and a second round accumulates on top."""
    from mvuld_amd import ops
    g_ = torch.Generator().manual_seed(rows + C)
    x = torch.randn(rows, C, generator=g_).to(torch.bfloat16).to(gpu)
    dy = torch.randn(rows, C, generator=g_).to(torch.bfloat16).to(gpu)
    gamma, beta = torch.nn.Parameter((torch.rand(C, generator=g_) + 0.5).to(gpu)), torch.nn.Parameter(torch.randn(C, generator=g_).to(gpu))
    _, mean, rstd, _ = ops.layernorm_fwd(x, gamma.data, beta.data, 1e-5)[:4]
    for q in (gamma, beta):
        q.grad = torch.zeros_like(q)
    dx0 = ops.layernorm_bwd(dy, x, gamma, beta, mean, rstd)                    # no group: reduction launched at once
    g0, b0 = gamma.grad.clone(), beta.grad.clone()
    assert float(g0.abs().max()) > 0
    main, wg = torch.cuda.current_stream(gpu), torch.cuda.Stream(device=gpu)
    assert ops.USE_LN_DEFER[0] and ops.WGRAD_STREAM[0] is None
    ops.WGRAD_STREAM[0] = (main.cuda_stream, wg)
    try:
        g2, b2 = torch.nn.Parameter(gamma.data.clone()), torch.nn.Parameter(beta.data.clone())      # a second LayerNorm in the same launch
        g2.grad, b2.grad = torch.zeros_like(g2), torch.zeros_like(b2)
        for rep in (1, 2):
            if rep == 1:
                gamma.grad.zero_(); beta.grad.zero_()
            dx1 = ops.layernorm_bwd(dy, x, gamma, beta, mean, rstd)
            dx2 = ops.layernorm_bwd(dx0, x, g2, b2, mean, rstd)
            assert len(ops._LN_PENDING) == 2
            torch.cuda.synchronize()
            assert rel(gamma.grad, (rep - 1) * g0) < 1e-6 if rep == 2 else float(gamma.grad.abs().max()) == 0.0      # not yet
            ops.join_wgrad_stream()
            assert not ops._LN_PENDING
            torch.cuda.synchronize()
            assert torch.equal(dx1, dx0)
            assert rel(gamma.grad, rep * g0) < 1e-6 and rel(beta.grad, rep * b0) < 1e-6
        ops.USE_LN_DEFER[0] = False
        try:
            r2, rb2 = torch.nn.Parameter(gamma.data.clone()), torch.nn.Parameter(beta.data.clone())
            r2.grad, rb2.grad = torch.zeros_like(r2), torch.zeros_like(rb2)
            assert torch.equal(ops.layernorm_bwd(dx0, x, r2, rb2, mean, rstd), dx2)
            assert rel(g2.grad, 2 * r2.grad) < 1e-6 and rel(b2.grad, 2 * rb2.grad) < 1e-6
        finally:
            ops.USE_LN_DEFER[0] = True
    finally:
        ops.WGRAD_STREAM[0] = None


@pytest.mark.parametrize("M,C", [(6272, 128), (1000, 128), (70000, 128)])
def test_fused_mlp_panel_kernels(gpu, M, C):
    """Round 3, csrc/mlp_panel.hip (Mlp.forward swin_transformer_v2.py:26-32 and its autograd at C = 128 / 256): the fused forward
    y = gelu(x W1^T + b1) W2^T + b2 (+ the activation, no pre-activation) and the fused backward dh = (dy W2) o gelu'(x W1^T + b1),
    dx = dh W1 + g against (a) fp32 torch on the bf16-rounded operands and (b) the three unfused GEMM launches they replace; row counts
    that are not multiples of the 128 / 256-row panels (clamped rows duplicate row M - 1), panels over several workgroup rounds."""
    from mvuld_amd import ops, hip
    g_ = torch.Generator().manual_seed(31 + C)
    bf = torch.bfloat16
    x = (torch.randn(M, C, generator=g_) * 0.8).to(bf)
    dy = torch.randn(M, C, generator=g_).to(bf)
    gres = torch.randn(M, C, generator=g_).to(bf)
    w1 = torch.nn.Parameter(torch.randn(4 * C, C, generator=g_) * C ** -0.5)
    b1 = torch.nn.Parameter(torch.randn(4 * C, generator=g_) * 0.3)
    w2 = torch.nn.Parameter(torch.randn(C, 4 * C, generator=g_) * (4 * C) ** -0.5)
    b2 = torch.nn.Parameter(torch.randn(C, generator=g_) * 0.3)
    W1, B1, W2, B2 = (torch.nn.Parameter(t.detach().to(gpu)) for t in (w1, b1, w2, b2))
    X, DY, G = x.to(gpu), dy.to(gpu), gres.to(gpu)
    assert bool(hip.LIB.fn("mvuld_mlp_fused_supported")(C))
    h, y = ops.mlp_fused_fwd(X, W1, B1, W2, B2)
    w1r, w2r = w1.detach().to(bf).float(), w2.detach().to(bf).float()
    pre = x.float() @ w1r.t() + b1.detach()
    h_ref = F.gelu(pre)
    y_ref = h_ref.to(bf).float() @ w2r.t() + b2.detach()
    assert rel(h, h_ref) < 1e-2 and rel(y, y_ref) < 1e-2, (rel(h, h_ref), rel(y, y_ref))
    _, y2 = ops.mlp_fused_fwd(X, W1, B1, W2, B2, need_h=False)
    assert torch.equal(y2, y)
    dh, dx = ops.mlp_fused_bwd(X, DY, G, W1, B1, W2)
    pr = pre.clone().requires_grad_(True)
    F.gelu(pr).sum().backward()
    dh_ref = (dy.float() @ w2r) * pr.grad
    dx_ref = dh_ref.to(bf).float() @ w1r + gres.float()
    assert rel(dh, dh_ref) < 1e-2 and rel(dx, dx_ref) < 1e-2, (rel(dh, dh_ref), rel(dx, dx_ref))
    _, dx0 = ops.mlp_fused_bwd(X, DY, None, W1, B1, W2)
    assert rel(dx0, dh_ref.to(bf).float() @ w1r) < 1e-2
    # the unfused launches
    hpre = torch.empty((M, 4 * C), dtype=bf, device=gpu)
    hu = ops.gemm_nt(X, ops.weight(W1, bf), bias=B1.data, epi=hip.EPI_GELU, aux=hpre)
    yu = ops.gemm_nt(hu, ops.weight(W2, bf), bias=B2.data)
    dhu = ops.gemm_nt(DY, ops.weight_t(W2, bf), epi=hip.EPI_MUL_DGELU, aux=hpre)
    dxu = ops.gemm_nt(dhu, ops.weight_t(W1, bf), epi=hip.EPI_ADD_AUX, aux=G)
    assert rel(h, hu) < 8e-3 and rel(y, yu) < 8e-3 and rel(dh, dhu) < 1.5e-2 and rel(dx, dxu) < 1.5e-2, (rel(h, hu), rel(y, yu), rel(dh, dhu), rel(dx, dxu))


@pytest.mark.parametrize("M,shapes", [(6272, [(512, 2048), (2048, 512), (512, 512), (1536, 512)]),          # a Swin stage-2 block (fc2, fc1, proj, qkv)
                                      (9917, [(768, 3072), (3072, 768), (768, 768), (2304, 768)]),          # a RoBERTa layer on a ragged packed token count
                                      (3136, [(1024, 4096), (4096, 1024), (384, 128), (1024, 1024), (3072, 1024)])])   # stage 3 + one ineligible product
def test_gemm_tn_wgrad_grouped(gpu, M, shapes):
    """Round 3: the weight gradients of one transformer block in ONE launch of the 256 x 256-tile kernel over a job table (+ one reduction
    launch), ops.wgrad_group / mvuld_gemm_tn_wgrad_group.  Every product must equal dY^T X (+ what the gradient buffer held) and its bias
    gradient colsum(dY); products the grouped kernel does not take (N x K far from filling 256 x 256 tiles) fall through to the
    single-product path inside the same group; a second group accumulates on top of the first; the split plan of a group differs from a
    lone product's, so equality with the ungrouped path is to fp32 summation order, not bit for bit."""
    from mvuld_amd import ops, hip
    prods = []
    for i, (N, K) in enumerate(shapes):
        dy, x = rt(T(f"gdy{i}", (M, N)), torch.bfloat16), rt(T(f"gx{i}", (M, K)), torch.bfloat16)
        w, b = torch.nn.Parameter(torch.zeros(N, K, device=gpu)), torch.nn.Parameter(torch.zeros(N, device=gpu))
        w.grad, b.grad = torch.ones(N, K, device=gpu), torch.ones(N, device=gpu)
        prods.append((dy, x, dev(dy, torch.bfloat16), dev(x, torch.bfloat16), w, b))
    assert ops.USE_WGRAD_GROUPS[0]
    taken = sum(int(hip.LIB.fn("mvuld_gemm_tn_wgrad_group_ok")(M, N, K, N, K)) for N, K in shapes)
    assert taken >= 4
    for rep in (1, 2):
        with ops.wgrad_group():
            for dy, x, gdy, gx, w, b in prods:
                ops.linear_wgrad(gdy, gx, w, b)
            assert len(ops._WGRAD_PENDING[0]) == taken          # deferred, not launched yet
        assert ops._WGRAD_PENDING[0] is None
        for dy, x, gdy, gx, w, b in prods:
            assert rel(w.grad, rep * (dy.t() @ x) + 1.0) < 2e-3
            assert rel(b.grad, rep * dy.sum(0) + 1.0) < 2e-3
    # the grouped launch shares the per-stream workspace with the 128 x 128 kernel's slab reduction, whose ticket block (the buffer's first
    # 4 KiB) must stay zero: a small weight gradient through that path right after a group must still be right
    sdy, sx = rt(T("gsdy", (392, 384)), torch.bfloat16), rt(T("gsx", (392, 128)), torch.bfloat16)
    sw, sb = torch.nn.Parameter(torch.zeros(384, 128, device=gpu)), torch.nn.Parameter(torch.zeros(384, device=gpu))
    sw.grad, sb.grad = torch.zeros(384, 128, device=gpu), torch.zeros(384, device=gpu)
    ops.linear_wgrad(dev(sdy, torch.bfloat16), dev(sx, torch.bfloat16), sw, sb)
    assert rel(sw.grad, sdy.t() @ sx) < 2e-3 and rel(sb.grad, sdy.sum(0)) < 2e-3
    # against the ungrouped path on the same operands
    ops.USE_WGRAD_GROUPS[0] = False
    try:
        for dy, x, gdy, gx, w, b in prods:
            g2 = w.grad.clone()
            w.grad.zero_(); b.grad.zero_()
            with ops.wgrad_group():
                ops.linear_wgrad(gdy, gx, w, b)
            assert rel(2.0 * w.grad + 1.0, g2) < 1e-5
    finally:
        ops.USE_WGRAD_GROUPS[0] = True


@pytest.mark.parametrize("B,h,w,S", [(2, 600, 800, 448), (1, 300, 200, 448), (3, 448, 448, 448), (1, 1000, 448, 448), (2, 97, 1301, 448),
                                     (1, 448, 900, 224)])
def test_image_ingest_matches_pillow_restatement(gpu, B, h, w, S):
    """mvuld_image_resize_bicubic_normalize (data/build.py:146-168 on the device) against the numpy restatement of Pillow's resampler
    (itself pinned byte for byte against PIL in the CPU suite): the resized 8-bit image must be IDENTICAL, the normalised float
    tensor equal to ToTensor + Normalize of it; batches, both passes, a skipped horizontal pass, enlarging, bf16 output."""
    from oracle import image_ref
    from mvuld_amd.data.image_ingest import DeviceImageTransform
    rng = np.random.default_rng(B * 1000 + h + w)
    imgs = rng.integers(0, 256, size=(B, h, w, 3), dtype=np.uint8)
    imgs[:, : h // 2] = (imgs[:, : h // 2].astype(np.int32) // 8 + np.linspace(0, 220, w)[None, None, :, None]).astype(np.uint8)
    tf = DeviceImageTransform(S)
    out, u8 = tf(torch.from_numpy(imgs).to(gpu), return_u8=True)
    for b in range(B):
        ref8 = image_ref.resize_bicubic_u8(imgs[b], S, S)
        assert np.array_equal(u8[b].cpu().numpy(), ref8), f"image {b}: resized bytes differ from Pillow's"
        want = image_ref.to_tensor_normalize(ref8)
        assert float(np.abs(out[b].cpu().numpy() - want).max()) <= 2.4e-7           # one ulp at |x| <= 2.7 (fp32 division rounding modes)
    out16 = DeviceImageTransform(S, out_dtype=torch.bfloat16)(torch.from_numpy(imgs[:1]).to(gpu))
    assert out16.dtype == torch.bfloat16 and float((out16.float() - out[:1]).abs().max()) <= 2.0 ** -7
    assert out.shape == (B, 3, S, S)


@pytest.mark.parametrize("sizes,deg", [([5, 1, 9], 3), ([200] * 32, 4), ([150, 250, 1, 77] * 64, 5)])
def test_graph_csr_index_on_device_equals_host(gpu, sizes, deg):
    """mvuld_graph_csr_build (stable radix sorts + gathers on the device) against the host builder BatchedGraph.index(): the six
    index arrays must be IDENTICAL -- including the edge order inside every destination / source group (edge-id order), which the
    GAT kernels' deterministic accumulation relies on.  Batches of 3, 32 and 256 graphs; isolated nodes; self loops; multi-edges."""
    from mvuld_amd.graph import BatchedGraph, add_self_loop
    gen = torch.Generator().manual_seed(len(sizes) * 31 + deg)
    srcs, dsts, off = [], [], 0
    for n in sizes:
        m = n * deg
        srcs.append(torch.randint(0, n, (m,), generator=gen) + off)
        dsts.append(torch.randint(0, max(1, n - 1), (m,), generator=gen) + off)        # the last node of each graph: no in-edges
        off += n
    g = add_self_loop(BatchedGraph(torch.cat(srcs), torch.cat(dsts), sizes))
    host = g.index()
    gd = BatchedGraph(g.src.to(gpu), g.dst.to(gpu), sizes)
    assert gd._index is None
    dev = gd.index()
    assert set(dev) == set(host)
    for k in host:
        assert dev[k].dtype == torch.int32 and dev[k].is_cuda and torch.equal(dev[k].cpu(), host[k]), k
