"""Thin, allocation-only wrappers over the C ABI (one Python function per kernel family).

PyTorch is used here for device memory (``torch.empty``), views and streams only; every
arithmetic operation is a ``libmvuld_hip.so`` call.  Tensors are contiguous; activations are
f32 or bf16, parameters / gradients / statistics fp32.
"""
import math
import os

import torch

from . import hip
from .hip import call, ptr, dt

# ---------------------------------------------------------------------------------------------
# weight views: bf16 working copies and transposed copies, refreshed when WEIGHT_EPOCH moves
WEIGHT_EPOCH = [0]
FORCE_SIMPLE_GEMM = [False]          # tests: route bf16 GEMMs through the VALU kernel
ATTN_IMPL = ["auto"]                 # "auto" | "simple"
USE_SPLIT3 = [False]                 # fp32 GEMMs (the head's fp32 tail) as 3-term bf16 splits on the matrix cores
SPLIT3_IN_REGISTERS = [os.environ.get("MVULD_SPLIT3_FUSED", "1") != "0"]      # ... split inside the GEMM kernel (0: two split launches + a 3K-deep product)
SPLIT3_TRANS = [os.environ.get("MVULD_SPLIT3_TRANS", "1") != "0"]             # ... Rs_GCN's R^T dY / dR ph products read their operands as stored (no transposes)
SPLIT3_WGRAD = [os.environ.get("MVULD_SPLIT3_WGRAD", "1") != "0"]             # ... the head's weight gradients as one such launch (0: two casts + the bf16 kernel)
# NOTE: with SPLIT3_WGRAD the token contraction of a head weight gradient is split over workgroups that add into .grad with
# fp32 atomics -- the order of those additions is not fixed, so the head's gradients (and a training run) are reproducible to fp32 rounding of
# the sum order, not bit for bit, for a fixed seed.  MVULD_SPLIT3_WGRAD=0 restores the ordered slab reduction (exact-equality tests use it).
USE_TN_WGRAD = [True]                # bf16 weight gradients through the transpose-free TN kernel
USE_TN_SLABS = [os.environ.get("MVULD_TN_SLABS", "1") != "0"]     # ... whose split contraction (2..8 ways) goes through a slab workspace, not atomics


# callbacks fired when the LAST backward of an encoder has run, i.e. all its gradients are final (used to start
# that encoder's gradient all-reduce while the rest of backward is still running)
_BACKWARD_DONE = {}


def on_backward_done(tag, fn, key="default"):
    """Register `fn` to run when the backward of encoder `tag` ("swin" | "unixcoder") has launched its last kernel (one
    callback per (tag, key); None removes it)."""
    d = _BACKWARD_DONE.setdefault(tag, {})
    if fn is None:
        d.pop(key, None)
    else:
        d[key] = fn


def fire_backward_done(tag):
    # a weight gradient still deferred in an open wgrad_group (the firing block's own, when its last product has no bias output to
    # close the group) must be in flight before anything that reads the gradients (the data-parallel exchange) is launched
    if _WGRAD_PENDING[0]:
        wgrad_group_flush()
    ln_reduce_flush()
    cbs = list(_BACKWARD_DONE.get(tag, {}).values())
    if (cbs or tag == "swin") and tag.startswith("swin"):
        join_wgrad_stream()               # the image encoder's weight gradients run on their own stream: finish them first
    for fn in cbs:
        fn()
    if tag == "swin":
        WGRAD_STREAM[0] = None            # the step's backward is over: later launches (other models, tests) stay on their own stream


def use_priority_main_stream(device=None):
    """Make the calling thread's current stream a HIGH-priority one (the training drivers call this once, after set_device).  The fused
    step's critical chain is the image encoder on the current stream; the text encoder, the graph branch and the weight gradients run on
    side streams created by the model at default priority.  With the chain's queue served first the step is 0.2-0.6 ms shorter (40-step
    A/B/A/B on two boxes: 51.56 / 51.53 -> 50.91 / 50.94 ms and 50.85 / 51.01 -> 50.63 / 50.38 ms; every side stream high and the chain low:
    +0.3 ms).  gfx950 has two levels (torch.cuda.Stream.priority_range() = (0, -1)).  MVULD_MAIN_PRIO=0 keeps the default stream."""
    prio = int(os.environ.get("MVULD_MAIN_PRIO", "-1"))
    if prio == 0 or not torch.cuda.is_available():
        return None
    s = torch.cuda.Stream(device=device, priority=prio)
    s.wait_stream(torch.cuda.current_stream(device))
    torch.cuda.set_stream(s)
    return s


def begin_step():
    """Called where a training step starts (FusedMVulD.forward): drops whatever a backward that raised left deferred -- LayerNorm
    parameter-gradient partials, an open weight-gradient group -- so that it cannot reach this step's gradients, and
    arms the zeroed-scratch pool."""
    del _LN_PENDING[:]
    if _WGRAD_PENDING[0]:
        del _WGRAD_PENDING[0][:]
    ZERO_POOL.begin_step()


class _ZeroPool:
    """Zeroed fp32 scratch of one training step's backward, carved from ONE buffer that one fill clears: the 24 Swin blocks each took
    their own torch.zeros before (table-gradient accumulator + q/v-bias column sums: 24 fills per step on the backward chain).
    `begin_step()` (the fused model's forward) retires the previous buffer and sizes the next one from what the last backward took;
    the first take() of a backward allocates and clears it on the CURRENT stream -- every later user, on any stream, is ordered behind
    kernels of this stream.  Without begin_step() (stand-alone modules, tests) every take() is its own torch.zeros, as before."""

    def __init__(self):
        self.buf, self.off, self.took, self.size, self.armed = None, 0, 0, 0, False

    def begin_step(self):
        self.size = max(self.size, self.took)
        self.buf, self.off, self.took, self.armed = None, 0, 0, True

    def take(self, n, device):
        n64 = (n + 63) // 64 * 64
        self.took += n64
        if self.armed and self.buf is None and self.size >= n64:
            self.buf, self.off = torch.zeros(self.size, dtype=torch.float32, device=device), 0
        if self.buf is not None and self.buf.device == device and self.off + n64 <= self.buf.numel():
            out = self.buf[self.off:self.off + n]
            self.off += n64
            return out
        return torch.zeros(n, dtype=torch.float32, device=device)


ZERO_POOL = _ZeroPool()


def defer_column_add(src, dst_lo=None, dst_hi=None, C=None, src_stream=None):
    """dst_lo += src[:C], dst_hi += src[C:2C] (fp32 vectors), deferred into the next batched reduction launch (ln_reduce_flush: it takes
    [nparts][2C] column partials, here nparts = 1) instead of one aten add_ per vector.  `src_stream`: the stream `src` was produced on (what linear_wgrad returns)."""
    if src_stream is None:
        src_stream = torch.cuda.current_stream(src.device)
    _LN_PENDING.append((src, 1, C, dst_lo, dst_hi, src_stream))


def bump_weight_epoch():
    WEIGHT_EPOCH[0] += 1


def refresh_store_transposes(store):
    """Refresh every registered transposed weight copy of a ParamStore in one launch per dtype (called right after the
    epoch moves: optimizer step / load_state_dict).  store._tjobs[dtype] = {"list": [(param, key, src, dst, R, C)], ...}."""
    for dtype, reg in getattr(store, "_tjobs", {}).items():
        if not reg["list"]:
            continue
        if reg["table"] is None:
            rows, t0 = [], 0
            for _, _, src, dst, R, C in reg["list"]:
                rows.append([src.data_ptr(), dst.data_ptr(), R, C, t0])
                t0 += math.ceil(R / 64) * math.ceil(C / 64)
            reg["table"] = torch.tensor(rows, dtype=torch.int64).to(reg["list"][0][2].device)
            reg["tiles"] = t0
        call("transpose_batched", ptr(reg["table"]), len(reg["list"]), reg["tiles"], hip.F32 if dtype == torch.float32 else hip.BF16)
        for p, key, *_ in reg["list"]:
            setattr(p, key + "_epoch", WEIGHT_EPOCH[0])


def _register_transpose(p, key, src, dst):
    jobs = p._mv_store.__dict__.setdefault("_tjobs", {})
    reg = jobs.setdefault(src.dtype, {"list": [], "table": None, "tiles": 0})
    reg["list"].append((p, key, src, dst, src.shape[0], src.shape[1]))
    reg["table"] = None


def weight(p: torch.Tensor, dtype) -> torch.Tensor:
    """Parameter as a GEMM operand of activation dtype `dtype` ([out, in], row-major)."""
    if dtype == torch.float32:
        return p.data if isinstance(p, torch.nn.Parameter) else p
    if getattr(p, "_mv_w16_store", False):       # ParamStore view: the AdamW kernel keeps it current
        return p._mv_w16
    c = getattr(p, "_mv_w16", None)
    if c is None or getattr(p, "_mv_w16_epoch", -1) != WEIGHT_EPOCH[0]:
        if c is None:
            c = torch.empty(p.shape, dtype=torch.bfloat16, device=p.device)
        call("cast", ptr(p), hip.F32, ptr(c), hip.BF16, p.numel())
        p._mv_w16 = c
        p._mv_w16_epoch = WEIGHT_EPOCH[0]
    return c


def weight_t(p: torch.Tensor, dtype) -> torch.Tensor:
    """Transposed operand [in, out] (for dgrad through the NT GEMM)."""
    key = "_mv_wt32" if dtype == torch.float32 else "_mv_wt16"
    c = getattr(p, key, None)
    if c is None or getattr(p, key + "_epoch", -1) != WEIGHT_EPOCH[0]:
        w = weight(p, dtype)
        w2 = w.reshape(w.shape[0], -1)
        if c is None:
            c = torch.empty((w2.shape[1], w2.shape[0]), dtype=w.dtype, device=w.device)
        call("transpose", ptr(w2), ptr(c), w2.shape[0], w2.shape[1], 1, dt(w2))
        fresh = getattr(p, key, None) is None
        setattr(p, key, c)
        setattr(p, key + "_epoch", WEIGHT_EPOCH[0])
        if fresh and getattr(p, "_mv_store", None) is not None:
            _register_transpose(p, key, w2, c)        # flat-store storage is stable: later refreshes are batched
    return c


def grad_of(p: torch.nn.Parameter) -> torch.Tensor:
    """fp32 gradient buffer of a parameter (atomic-accumulate target)."""
    if p.grad is None:
        p.grad = torch.zeros_like(p.data, dtype=torch.float32)
    st = getattr(p, "_mv_store", None)
    if st is not None:
        st.grads_zeroed = False           # something is about to accumulate: the next zero_grad() must really clear
    return p.grad


# ---------------------------------------------------------------------------------------------
def f32x3_ok(dtype, M, N):
    """fp32 products of this size run as in-register bf16 hi / lo splits (mvuld_gemm_nt_f32x3): the only route that takes the ta / tb
    (transposed-storage operand) arguments of gemm_nt."""
    return dtype == torch.float32 and USE_SPLIT3[0] and SPLIT3_IN_REGISTERS[0] and not FORCE_SIMPLE_GEMM[0] and M >= 64 and N >= 64


def gemm_nt(a, b, out=None, bias=None, epi=hip.EPI_NONE, aux=None, alpha=1.0, out_mode=hip.OUT_STORE, splitk=1,
            out_dtype=None, M=None, N=None, K=None, lda=None, ldb=None, ldc=None, batch=1, sa=0, sb=0, sc=0,
            ldaux=None, saux=0, ta=False, tb=False):
    """out[M,N] = epi(alpha * a[M,K] @ b[N,K]^T + bias).  2-D contiguous by default; explicit ld/stride for views.
    ta / tb (f32x3_ok products only; pass M, N, K and the leading dimensions): that operand is stored [K, M] / [K, N]."""
    M = a.shape[-2] if M is None else M
    K = a.shape[-1] if K is None else K
    N = b.shape[-2] if N is None else N
    lda = a.stride(-2) if lda is None else lda
    ldb = b.stride(-2) if ldb is None else ldb
    if out is None:
        od = out_dtype or a.dtype
        shape = (M, N) if batch == 1 else (batch, M, N)
        out = torch.empty(shape, dtype=od, device=a.device)
        if batch > 1 and sc == 0:
            sc = M * N
    ldc = out.stride(-2) if ldc is None else ldc
    if aux is not None and ldaux is None:
        ldaux = aux.stride(-2)
    if bias is not None and bias.dtype != torch.float32:
        raise TypeError("gemm_nt: bias must be fp32")
    if (a.dtype == torch.float32 and b.dtype == torch.float32 and USE_SPLIT3[0] and not FORCE_SIMPLE_GEMM[0] and M >= 64 and N >= 64
            and (aux is None or aux.dtype == out.dtype)):
        if (SPLIT3_IN_REGISTERS[0] and (splitk == 1 or out_mode == hip.OUT_ATOMIC) and out.dtype == torch.float32
                and (out_mode != hip.OUT_ATOMIC or epi <= hip.EPI_BIAS)):
            # ... split on the way from memory to LDS: one launch, no operand copies (mvuld_gemm_nt_f32x3)
            if hip.TIMING.enabled:
                hip.TIMING.annotate("gemm_nt_mfma_bf16(split3 fp32)", 6.0 * M * N * K * batch)
            call("gemm_nt_f32x3", ptr(a), lda, sa, ptr(b), ldb, sb, ptr(out), ldc, sc, M, N, K, batch, ptr(bias), epi, ptr(aux), ldaux or 0, saux,
                 float(alpha), out_mode, 1 if ta else 0, 1 if tb else 0, splitk)
            return out
        if ta or tb:
            raise RuntimeError("gemm_nt: transposed-storage operands exist on the in-register split route only (ops.f32x3_ok)")
        # near-fp32 product on the bf16 matrix cores: both operands split into hi/lo bf16 parts concatenated along K
        Kp = (K + 7) // 8 * 8
        ra, rb = M * batch, N * batch
        if (batch == 1 or (sa == M * lda and sb == N * ldb)):
            a3 = torch.empty((ra, 3 * Kp), dtype=torch.bfloat16, device=a.device)
            b3 = torch.empty((rb, 3 * Kp), dtype=torch.bfloat16, device=a.device)
            call("split_bf16x3", ptr(a), lda, ptr(a3), ra, K, Kp, 0)
            call("split_bf16x3", ptr(b), ldb, ptr(b3), rb, K, Kp, 1)
            if hip.TIMING.enabled:
                hip.TIMING.annotate("gemm_nt_mfma_bf16(split3 fp32)", 6.0 * M * N * Kp * batch)
            call("gemm_nt", ptr(a3), 3 * Kp, M * 3 * Kp, ptr(b3), 3 * Kp, N * 3 * Kp, ptr(out), ldc, sc, M, N, 3 * Kp, batch, ptr(bias), epi,
                 ptr(aux), ldaux or 0, saux, float(alpha), out_mode, splitk, hip.BF16, dt(out), 0)
            return out
    if ta or tb:
        raise RuntimeError("gemm_nt: transposed-storage operands exist on the in-register split route only (ops.f32x3_ok)")
    if hip.TIMING.enabled:
        mfma = (a.dtype == torch.bfloat16 and K % 8 == 0 and lda % 8 == 0 and ldb % 8 == 0 and sa % 8 == 0 and sb % 8 == 0
                and M >= 32 and N >= 32 and not FORCE_SIMPLE_GEMM[0])
        hip.TIMING.annotate("gemm_nt_mfma_bf16" if mfma else "gemm_nt_simple", 2.0 * M * N * K * batch)
    call("gemm_nt", ptr(a), lda, sa, ptr(b), ldb, sb, ptr(out), ldc, sc, M, N, K, batch, ptr(bias), epi, ptr(aux),
         ldaux or 0, saux, float(alpha), out_mode, splitk, dt(a), dt(out), 1 if FORCE_SIMPLE_GEMM[0] else 0)
    return out


# --------------------------------------------------------------------------------------------- fp8 forward GEMMs (BASELINE configs[4])
FP8_IN_FUSED = [False]             # set by the fused model while its forward runs: it rolls the fp8 scales once for both encoders
FP8_FWD = [False]                    # forward QKV / FFN products of the two encoders on the fp8 matrix cores
_Q_SCRATCH = {}


def fp8_eligible(M, N, K):
    return K % 64 == 0 and K >= 256 and N % 8 == 0 and M >= 256


def quant_fp8(x, out=None, scale=None):
    """Per-tensor OCP e4m3 quantisation of a contiguous bf16 / fp32 tensor: (q uint8 same shape, scale fp32[1]), x ~= q * scale."""
    assert x.is_contiguous() and x.numel() % 8 == 0
    key = (x.device, torch.cuda.current_stream().cuda_stream if x.is_cuda else 0)
    part = _Q_SCRATCH.get(key)
    if part is None:
        part = _Q_SCRATCH[key] = torch.empty(1024, dtype=torch.float32, device=x.device)
    q = torch.empty(x.shape, dtype=torch.uint8, device=x.device) if out is None else out
    sc = torch.empty(1, dtype=torch.float32, device=x.device) if scale is None else scale
    call("quant_e4m3", ptr(x), x.numel(), dt(x), ptr(q), ptr(sc), ptr(part))
    return q, sc


def weight_fp8(p):
    """(e4m3 copy, scale) of a weight [out, in], requantised when the optimizer has stepped (WEIGHT_EPOCH).  Weights that live in a
    ParamStore are registered on first use and from then on refreshed all together, in three launches, right after the optimizer
    step (refresh_store_fp8); anything else is requantised here, lazily."""
    c = getattr(p, "_mv_w8", None)
    if c is None or p._mv_w8_epoch != WEIGHT_EPOCH[0]:
        w = p.data if isinstance(p, torch.nn.Parameter) else p
        q, sc = quant_fp8(w.reshape(w.shape[0], -1), *(c or (None, None)))
        fresh = c is None
        p._mv_w8 = c = (q, sc)
        p._mv_w8_epoch = WEIGHT_EPOCH[0]
        st = getattr(p, "_mv_store", None)
        if fresh and st is not None and w.dtype == torch.float32 and w.is_contiguous():
            reg = st.__dict__.setdefault("_q8jobs", {"list": [], "table": None, "blocks": 0, "partials": None})
            reg["list"].append((p, w, q, sc))
            reg["table"] = None
    return c


def register_qkv_bias(q_bias, v_bias, dst, owner):
    """A SwinV2 block's packed (q_bias, 0, v_bias) buffer: when the two parameters live in a ParamStore the buffer joins the
    store's job table and refresh_store_qkv_bias() rebuilds all of them in one launch after each optimizer step."""
    st = getattr(q_bias, "_mv_store", None)
    if st is None or getattr(v_bias, "_mv_store", None) is not st or not dst.is_cuda:
        return
    reg = st.__dict__.setdefault("_qbjobs", {"list": [], "table": None})
    reg["list"].append((q_bias, v_bias, dst, owner))
    reg["table"] = None


def refresh_store_qkv_bias(store):
    reg = getattr(store, "_qbjobs", None)
    if not reg or not reg["list"]:
        return
    live = [(q, v, d, o) for q, v, d, o in reg["list"] if o._qb is d]          # a block that re-made its buffer re-registers it
    if len(live) != len(reg["list"]):
        reg["list"], reg["table"] = live, None
        if not live:
            return
    if reg["table"] is None:
        rows = [[q.data.data_ptr(), v.data.data_ptr(), d.data_ptr(), q.numel()] for q, v, d, _ in reg["list"]]
        reg["table"] = torch.tensor(rows, dtype=torch.int64).to(reg["list"][0][2].device)
    call("qkv_bias_pack_batched", ptr(reg["table"]), len(reg["list"]))
    for _, _, _, o in reg["list"]:
        o._qb_epoch = WEIGHT_EPOCH[0]


def register_cpb(attn, hidden, table16, owner):
    """A SwinV2 block's continuous-position-bias buffers join the store's job table (see refresh_store_cpb)."""
    ps = (attn.cpb_mlp[0].weight, attn.cpb_mlp[0].bias, attn.cpb_mlp[2].weight)
    st = getattr(ps[0], "_mv_store", None)
    if st is None or any(getattr(p, "_mv_store", None) is not st for p in ps) or not hidden.is_cuda:
        return
    reg = st.__dict__.setdefault("_cpbjobs", {"list": [], "table": None, "rows": 0})
    reg["list"].append((attn, hidden, table16, owner))
    reg["table"] = None


def refresh_store_cpb(store):
    """All registered bias tables in one launch, right after the weight epoch moved."""
    reg = getattr(store, "_cpbjobs", None)
    if not reg or not reg["list"]:
        return
    live = [j for j in reg["list"] if getattr(j[3], "_cpb", None) is not None and j[3]._cpb[0] is j[1]]
    if len(live) != len(reg["list"]):
        reg["list"], reg["table"] = live, None
        if not live:
            return
    if reg["table"] is None:
        rows, r0 = [], 0
        for a, hid, tab, _ in reg["list"]:
            T2, H = tab.shape
            rows.append([a.relative_coords_table.data_ptr(), a.cpb_mlp[0].weight.data.data_ptr(), a.cpb_mlp[0].bias.data.data_ptr(),
                         a.cpb_mlp[2].weight.data.data_ptr(), hid.data_ptr(), tab.data_ptr(), T2, H, r0])
            r0 += T2
        reg["table"] = torch.tensor(rows, dtype=torch.int64).to(reg["list"][0][1].device)
        reg["rows"] = r0
    call("cpb_table_fwd_batched", ptr(reg["table"]), len(reg["list"]), reg["rows"])
    for _, _, _, o in reg["list"]:
        o._cpb_epoch = WEIGHT_EPOCH[0]


def refresh_store_fp8(store):
    """Requantise every registered fp8 weight copy of a ParamStore (called right after the epoch moves, like the transposes)."""
    reg = getattr(store, "_q8jobs", None)
    if not reg or not reg["list"] or not FP8_FWD[0]:
        return
    if reg["table"] is None:
        rows, b0 = [], 0
        for _, w, q, sc in reg["list"]:
            rows.append([w.data_ptr(), q.data_ptr(), sc.data_ptr(), w.numel(), b0])
            b0 += (w.numel() + 8191) // 8192
        dev = reg["list"][0][1].device
        reg["table"] = torch.tensor(rows, dtype=torch.int64).to(dev)
        reg["blocks"] = b0
        reg["partials"] = torch.empty(b0, dtype=torch.float32, device=dev)
    call("quant_e4m3_batched", ptr(reg["table"]), len(reg["list"]), reg["blocks"], ptr(reg["partials"]))
    for p, *_ in reg["list"]:
        p._mv_w8_epoch = WEIGHT_EPOCH[0]


class Fp8Site:
    """One fused quantisation site: a tensor some kernel produces (a LayerNorm output, a GELU activation) and the next fp8 product
    consumes.  Delayed scaling: the producer writes the e4m3 copy under the scale of the PREVIOUS step while folding max|value| into
    the site's amax; fp8_roll() turns that into the next step's scale.  `state` = the {scale, amax} pair in device memory.  A site
    that has never seen data (`cal` False) is calibrated by one dynamic two-pass quantisation, which leaves its scale in the slot."""
    __slots__ = ("state", "cal")

    def __init__(self, state):
        self.state, self.cal = state, False


_FP8_STATE = {}      # device -> [flat fp32 tensor of {scale, amax} pairs, sites handed out]
_FP8_MAX_SITES = 4096
_FP8_SIDE = {}       # data_ptr of a bf16 activation -> (q, scale) emitted beside it for the ONE consumer about to read it


def fp8_site(owner, name, device):
    key = "_mv_q8_" + name
    st = getattr(owner, key, None)
    if st is None or st.state.device != device:
        reg = _FP8_STATE.get(device)
        if reg is None:
            reg = _FP8_STATE[device] = [torch.zeros(2 * _FP8_MAX_SITES, dtype=torch.float32, device=device), 0]
        assert reg[1] < _FP8_MAX_SITES
        st = Fp8Site(reg[0][2 * reg[1]:2 * reg[1] + 2])
        reg[1] += 1
        setattr(owner, key, st)
    return st


def fp8_roll(device):
    """Start of a forward pass: every site's amax of the last pass becomes its scale (one launch for all sites)."""
    _FP8_SIDE.clear()
    reg = _FP8_STATE.get(device)
    if reg is not None and reg[1] > 0:
        call("fp8_roll_scales", ptr(reg[0]), reg[1])


def fp8_put(x, qpair):
    if qpair is not None:
        _FP8_SIDE[x.data_ptr()] = (qpair, x.shape)


def fp8_take(x):
    e = _FP8_SIDE.pop(x.data_ptr(), None)
    return e[0] if e is not None and e[1] == x.shape else None


def gemm_nt_fp8(a8, sa, b8, sb, bias=None, epi=hip.EPI_NONE, aux=None, out=None, emit=None, need_out=True):
    """out[M,N] (bf16) = epi(sa * sb * a8 . b8^T + bias).  emit = Fp8Site: the GELU epilogue also writes the activation as e4m3 under
    the site's scale -> returns (out, (q, scale)); need_out False (inference) then skips the bf16 copy (out = None)."""
    M, K = a8.shape
    N = b8.shape[0]
    if out is None and (need_out or emit is None):
        out = torch.empty((M, N), dtype=torch.bfloat16, device=a8.device)
    q = torch.empty((M, N), dtype=torch.uint8, device=a8.device) if emit is not None else None
    if hip.TIMING.enabled:
        hip.TIMING.annotate("gemm_nt_mfma_fp8", 2.0 * M * N * K)
    call("gemm_nt_fp8", ptr(a8), a8.stride(0), ptr(b8), b8.stride(0), ptr(out), out.stride(0) if out is not None else N, M, N, K, ptr(bias),
         epi, ptr(aux), aux.stride(0) if aux is not None else 0, ptr(sa), ptr(sb), ptr(q), N, ptr(emit.state) if emit is not None else None)
    if emit is not None:
        return out, (q, emit.state[0:1])
    return out


def linear_fwd(x, w_param, bias=None, epi=hip.EPI_NONE, aux=None, xq=None, emit=None, need_out=True):
    """Forward product of a Linear layer, x[M,K] . W[N,K]^T (+ bias, GELU): on the fp8 matrix cores when FP8_FWD is on and the shape
    is eligible, else the bf16 / fp32 GEMM.  Weights are quantised once per optimizer step.  The activation operand comes either
    pre-quantised from its producer (xq = (q, scale): see Fp8Site) or through the dynamic two-pass quantisation here.
    emit = Fp8Site of the OUTPUT (GELU products only): returns (out, qpair-or-None) instead of out."""
    src = xq[0] if xq is not None else x
    if (FP8_FWD[0] and (xq is not None or (x.dtype == torch.bfloat16 and x.is_contiguous()))
            and fp8_eligible(src.shape[0], w_param.shape[0], src.shape[1])):
        q, sc = xq if xq is not None else quant_fp8(x)
        w8, ws = weight_fp8(w_param)
        if emit is not None and not emit.cal:
            # first pass through this site: bf16 out now, and its dynamic quantisation (by the consumer) calibrates the slot
            out = gemm_nt_fp8(q, sc, w8, ws, bias=bias, epi=epi, aux=aux)
            qo, _ = quant_fp8(out, scale=emit.state[0:1])
            emit.cal = True
            return out, (qo, emit.state[0:1])
        return gemm_nt_fp8(q, sc, w8, ws, bias=bias, epi=epi, aux=aux, emit=emit, need_out=need_out)
    out = gemm_nt(x, weight(w_param, x.dtype), bias=bias, epi=epi, aux=aux)
    return (out, None) if emit is not None else out


# The FFN's training pair (csrc/gemm_common.h): the forward product leaves gelu'(pre-activation) in the buffer that used to hold the
# pre-activation, the backward product of the second Linear multiplies by it instead of recomputing erf + exp per element.
USE_GELU_DG = [os.environ.get("MVULD_GELU_DG", "1") != "0"]


def gelu_epi(aux):
    """Epilogue code of the forward fc1 / intermediate.dense product given its side buffer (None: inference)."""
    return hip.EPI_GELU_DG if (aux is not None and USE_GELU_DG[0]) else hip.EPI_GELU


def dgelu_epi(aux):
    """... and of the matching backward product: decided in the forward and carried in the autograd context."""
    return hip.EPI_MUL_AUX if (aux is not None and USE_GELU_DG[0]) else hip.EPI_MUL_DGELU


# fused MLP (csrc/mlp_panel.hip): widest block width that takes it -- 0 = never, 128 (default) = Swin stage 0 (the library builds C = 128 only)
FUSED_MLP_MAX_C = [int(os.environ.get("MVULD_FUSED_MLP", "128"))]
FUSED_MLP_TRAIN = [os.environ.get("MVULD_FUSED_MLP_TRAIN", "1") != "0"]      # also in training (forward + recomputing backward)


def mlp_fused_ok(x, fc1_w, training=False):
    """The fused MLP kernels take bf16 blocks of width 128 / 256 with hidden = 4C (Swin stages 0 and 1).  Measured (tools/bench_mlp.py,
    DESIGN section 9c): the forward wins at C = 128 (263 vs 363 us, less without the activation copy: inference), the recomputing backward
    about breaks even (427 vs 403 us) and C = 256 loses on both (259 vs 215, 368 vs 215 us) -- the default is C = 128 only, in training
    (step 55.63 -> 55.29 ms) and inference (batch 256: 113.3 -> 111.5 ms)."""
    C = x.shape[1]
    if training and not FUSED_MLP_TRAIN[0]:
        return False
    return (C <= FUSED_MLP_MAX_C[0] and x.is_cuda and x.dtype == torch.bfloat16 and not FORCE_SIMPLE_GEMM[0] and x.is_contiguous()
            and tuple(fc1_w.shape) == (4 * C, C) and bool(hip.LIB.fn("mvuld_mlp_fused_supported")(C)))


def mlp_fused_fwd(x, fc1_w, fc1_b, fc2_w, fc2_b, need_h=True):
    """-> (h = gelu(x W1^T + b1) [M, 4C] (None without need_h), y = h W2^T + b2 [M, C]); the pre-activation is never written."""
    M, C = x.shape
    h = torch.empty((M, 4 * C), dtype=x.dtype, device=x.device) if need_h else None
    y = torch.empty((M, C), dtype=x.dtype, device=x.device)
    hip.TIMING.annotate("gemm_nt_mfma_bf16", 2.0 * 2.0 * M * C * 4 * C)
    call("mlp_fused_fwd", ptr(x), ptr(weight(fc1_w, x.dtype)), ptr(fc1_b.data), ptr(weight(fc2_w, x.dtype)), ptr(fc2_b.data), ptr(h), ptr(y), M, C)
    return h, y


def mlp_fused_bwd(x, dy, g, fc1_w, fc1_b, fc2_w):
    """-> (dh = (dy W2) o gelu'(x W1^T + b1) [M, 4C], dx = dh W1 + g [M, C])."""
    M, C = x.shape
    dh = torch.empty((M, 4 * C), dtype=x.dtype, device=x.device)
    dx = torch.empty((M, C), dtype=x.dtype, device=x.device)
    hip.TIMING.annotate("gemm_nt_mfma_bf16", 3.0 * 2.0 * M * C * 4 * C)
    call("mlp_fused_bwd", ptr(x), ptr(dy), ptr(g), ptr(weight(fc1_w, x.dtype)), ptr(fc1_b.data), ptr(weight_t(fc2_w, x.dtype)),
         ptr(weight_t(fc1_w, x.dtype)), ptr(dh), ptr(dx), M, C)
    return dh, dx


def transpose(x, R=None, C=None, batch=1, out=None):
    R = x.shape[-2] if R is None else R
    C = x.shape[-1] if C is None else C
    if out is None:
        out = torch.empty((C, R) if batch == 1 else (batch, C, R), dtype=x.dtype, device=x.device)
    call("transpose", ptr(x), ptr(out), R, C, batch, dt(x))
    return out


def colsum_into(x, out, M=None, N=None, ld=None, col0=0):
    """out[0:N] += column sums of x[:, col0:col0+N]."""
    M = x.shape[0] if M is None else M
    N = x.shape[1] if N is None else N
    ld = x.stride(0) if ld is None else ld
    call("colsum", x.data_ptr() + col0 * x.element_size(), ld, ptr(out), M, N, dt(x))


def wgrad_splitk(n_out, k_out, depth):
    tiles = math.ceil(n_out / 128) * math.ceil(k_out / 128)
    ktiles = max(1, depth // 64)
    want = max(1, 768 // tiles)
    return int(max(1, min(want, ktiles // 4 if ktiles >= 8 else 1)))


# Weight-gradient stream: (main stream handle, side stream) set by the fused model for the duration of a step; weight gradients
# issued from the main stream are moved to it (the other branches already run beside the main stream).
WGRAD_STREAM = [None]


def wgrad_stream_for_current():
    ws = WGRAD_STREAM[0]
    if ws is None or hip.TIMING.enabled:
        return None
    main, wg = ws
    return wg if torch.cuda.current_stream().cuda_stream == main else None


def join_wgrad_stream():
    """Make the current stream wait for every weight gradient issued so far (before anything reads them: all-reduce, optimizer)."""
    ln_reduce_flush()
    ws = WGRAD_STREAM[0]
    if ws is not None:
        ev = torch.cuda.Event()
        ev.record(ws[1])
        torch.cuda.current_stream().wait_event(ev)


_TN_WS = {}


def _tn_workspace(device, nbytes):
    """Zero-initialised split-K exchange buffer of the weight-gradient kernel, one per (device, stream): its ticket words must be
    0 before a launch and the kernel leaves them 0, so it is cleared once, here, and then only ever reused in stream order."""
    if nbytes <= 0:
        return None
    key = (device.type, device.index, torch.cuda.current_stream(device).cuda_stream)
    ws = _TN_WS.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = _TN_WS[key] = torch.zeros(max(nbytes, 64 << 20), dtype=torch.uint8, device=device)
    return ws


# Grouped weight gradients (round 3): inside `with wgrad_group():` -- one transformer block's backward -- the eligible bf16 weight
# gradients are collected and issued as ONE launch of the 256 x 256-tile kernel over a job table (+ one reduction launch) when the
# group closes: together a block's products fill the chip with ~5-way instead of 16-21-way contraction splits, i.e. a third of the
# fp32 partial-slab traffic (csrc/gemm_tn256.hip).  Nothing in backward reads a weight gradient, so deferring them to the end of the
# block changes no dependency; operands stay referenced by the pending list until the launch.
USE_WGRAD_GROUPS = [os.environ.get("MVULD_WGRAD_GROUP", "1") != "0"]
_WGRAD_PENDING = [None]
# LayerNorm parameter gradients, deferred further (layernorm_bwd): (partials, nparts, C, dgamma, dbeta, producing stream).  While the image encoder's weight
# gradients go to the weight-gradient stream, a LayerNorm backward writes its column partials into a buffer of its own and the reductions
# that add them into the gradients are issued TOGETHER, one launch for the whole encoder (or stage, under data parallelism), when its
# backward-done hook fires.  Measured (bench.py, 60 steps): the 48 per-LayerNorm reduction launches cost the step 1.7-2.6 ms although they
# run 5 us each alone -- their 1024-thread workgroups found no CU with room beside the persistent GEMMs of the other streams.
USE_LN_DEFER = [os.environ.get("MVULD_LN_DEFER", "1") != "0"]
_LN_PENDING = []


def ln_reduce_flush():
    """Issue the pending LayerNorm parameter-gradient reductions in one launch (on the weight-gradient stream when one is active)."""
    if not _LN_PENDING:
        return
    lns = list(_LN_PENDING)
    del _LN_PENDING[:]
    import ctypes
    dev = lns[0][0].device
    wg = wgrad_stream_for_current()
    dst = wg if wg is not None else torch.cuda.current_stream(dev)
    seen = set()
    for part, nparts, C, dg, db, src in lns:               # order the launch behind every stream that produced partials
        if src.cuda_stream != dst.cuda_stream and src.cuda_stream not in seen:
            seen.add(src.cuda_stream)
            ev = torch.cuda.Event()
            ev.record(src)
            dst.wait_event(ev)
        if src.cuda_stream != dst.cuda_stream:
            part.record_stream(dst)
    desc = (ctypes.c_int64 * (5 * len(lns)))()
    for k, (part, nparts, C, dg, db, src) in enumerate(lns):
        desc[5 * k:5 * k + 5] = [part.data_ptr(), nparts, C, dg.data_ptr() if dg is not None else 0, db.data_ptr() if db is not None else 0]
    with torch.cuda.stream(dst):
        call("layernorm_bwd_reduce_batch", ctypes.addressof(desc), len(lns))


class wgrad_group:
    def __enter__(self):
        self.outer = _WGRAD_PENDING[0]
        if USE_WGRAD_GROUPS[0] and self.outer is None:
            _WGRAD_PENDING[0] = []
        return self

    def __exit__(self, *exc):
        if self.outer is None and _WGRAD_PENDING[0] is not None:
            try:
                if exc[0] is None:
                    wgrad_group_flush()
                else:
                    del _LN_PENDING[:]        # a backward that raised: its deferred LayerNorm partials must not reach a later step's gradients
            finally:
                _WGRAD_PENDING[0] = None
        return False


def wgrad_group_flush():
    """Issue the pending weight gradients (on the weight-gradient stream when one is active); returns the stream they went to."""
    jobs = _WGRAD_PENDING[0]
    dev = jobs[0][0].device if jobs else None
    if not jobs:
        return None
    _WGRAD_PENDING[0] = []
    import ctypes
    wg = wgrad_stream_for_current()
    cur = torch.cuda.current_stream(dev)
    if wg is not None:
        ev = torch.cuda.Event()
        ev.record(cur)
        wg.wait_event(ev)
        for dy, x, gw, bias_dst, M, N, K in jobs:
            dy.record_stream(wg)
            x.record_stream(wg)
            if bias_dst is not None:
                bias_dst.record_stream(wg)
    with torch.cuda.stream(wg if wg is not None else cur):
        for i in range(0, len(jobs), 8):
            chunk = jobs[i:i + 8]
            desc = (ctypes.c_int64 * (10 * len(chunk)))()
            flops = 0.0
            for k, (dy, x, gw, bias_dst, M, N, K) in enumerate(chunk):
                desc[10 * k:10 * k + 10] = [dy.data_ptr(), dy.stride(0), x.data_ptr(), x.stride(0), gw.data_ptr(), K, M, N, K,
                                            bias_dst.data_ptr() if bias_dst is not None else 0]
                flops += 2.0 * M * N * K
            need = hip.LIB.fn("mvuld_gemm_tn_wgrad_group_workspace_bytes")(ctypes.addressof(desc), len(chunk))
            if need < 0:
                raise RuntimeError("mvuld_gemm_tn_wgrad_group: ineligible product in a group")
            ws = _tn_workspace(dev, need)
            hip.TIMING.annotate("gemm_tn_wgrad", flops)
            call("gemm_tn_wgrad_group", ctypes.addressof(desc), len(chunk), ptr(ws), ws.numel())
    return wg if wg is not None else cur


def _tn_wgrad_call(dy, x, gw, bias_dst, M, N, K, splitk):
    ws = _tn_workspace(dy.device, hip.LIB.fn("mvuld_gemm_tn_wgrad_workspace_bytes")(M, N, K, splitk)) if USE_TN_SLABS[0] else None
    call("gemm_tn_wgrad", ptr(dy), dy.stride(0), ptr(x), x.stride(0), ptr(gw), K, M, N, K, ptr(bias_dst), splitk,
         ptr(ws), ws.numel() if ws is not None else 0)


# Streams other than the caller's that accumulate parameter gradients during a step (the fused model's side and weight-gradient
# streams).  Gradients are written by raw atomics into the flat store, invisible to autograd's stream bookkeeping, so everything
# that READS the gradient buffer (all-reduce launch, norm, AdamW) first makes its stream wait for all of them -- cheap, and
# independent of the order in which autograd happened to replay the branches.
GRAD_STREAMS = []


def register_grad_stream(stream):
    if all(s.cuda_stream != stream.cuda_stream for s in GRAD_STREAMS):
        GRAD_STREAMS.append(stream)


def join_grad_streams():
    cur = torch.cuda.current_stream()
    for s in GRAD_STREAMS:
        if s.cuda_stream != cur.cuda_stream:
            ev = torch.cuda.Event()
            ev.record(s)
            cur.wait_event(ev)


def linear_wgrad(dy, x, w_param, b_param=None, dyT=None, xT=None, bias_out=None):
    """dW[N,K] += dy[M,N]^T @ x[M,K]  and  db[N] += colsum(dy), into the parameters' fp32 .grad buffers.
    bias_out: an fp32 [N] buffer to accumulate colsum(dy) into instead of b_param.grad.  Returns the stream the gradient
    kernels were issued on (the weight-gradient stream when one is active, else the current one): whoever consumes
    bias_out must do so on that stream."""
    M, N = dy.shape
    K = x.shape[1]
    if (w_param is not None and dy.dtype == torch.float32 and x.dtype == torch.float32 and USE_TN_WGRAD[0] and dyT is None and SPLIT3_WGRAD[0] and f32x3_ok(torch.float32, N, K)
            and M >= 64 and dy.stride(1) == 1 and x.stride(1) == 1):
        # fp32 tail of a bf16 model, one launch: dW += dy^T x with both operands read as they are stored ([M, N] = the product's [K, M] operand,
        # [M, K] = its [K, N] operand), split into bf16 hi / lo parts inside the kernel (near-fp32: better than the bf16-rounded operands of the
        # route below, which also cost two casts), the token contraction split over workgroups that add into the gradient with atomics
        if bias_out is not None:
            colsum_into(dy, bias_out)
        elif b_param is not None:
            colsum_into(dy, grad_of(b_param))
        tiles = math.ceil(N / 64) * math.ceil(K / 64)
        splitk = max(1, min(math.ceil(M / 256), 512 // max(1, tiles)))
        hip.TIMING.annotate("gemm_tn_wgrad", 2.0 * M * N * K)
        gemm_nt(dy, x, out=grad_of(w_param).view(N, K), M=N, N=K, K=M, lda=dy.stride(0), ldb=x.stride(0), ldc=K, ta=True, tb=True,
                out_mode=hip.OUT_ATOMIC, splitk=splitk)
        return torch.cuda.current_stream(dy.device)
    if (w_param is not None and dy.dtype == torch.float32 and x.dtype == torch.float32 and USE_SPLIT3[0] and USE_TN_WGRAD[0]
            and not FORCE_SIMPLE_GEMM[0] and N % 8 == 0 and K % 8 == 0 and M >= 64 and dy.is_contiguous() and x.is_contiguous()):
        # fp32 tail of a bf16 model: its weight gradients are formed like every other weight gradient of the model, from
        # bf16-rounded operands with fp32 accumulation, through the transpose-free kernel (the fp32 route costs two transposes,
        # two 3-term splits and a 3x longer product per weight)
        # (the bias gradient stays an fp32 column sum: behind a BatchNorm it is a sum of terms that cancel to ~0)
        if bias_out is not None:
            colsum_into(dy, bias_out)
        elif b_param is not None:
            colsum_into(dy, grad_of(b_param))
        bias_out, b_param = None, None
        dy, x, dyT, xT = cast(dy, torch.bfloat16), cast(x, torch.bfloat16), None, None
    if (w_param is not None and dy.dtype == torch.bfloat16 and N % 8 == 0 and K % 8 == 0 and dyT is None and not FORCE_SIMPLE_GEMM[0]
            and USE_TN_WGRAD[0]):
        tiles = math.ceil(N / 128) * math.ceil(K / 128)
        # Split the token contraction so that tiles x splitk fills whole rounds of the 512 workgroup slots (2 per CU): one
        # round when it fills >= 90 % of them, else two; more splits only add fp32 atomics (measured: tools/bench_gemm.py).
        sk1, sk2 = max(1, 512 // tiles), max(1, 1024 // tiles)
        f1, f2 = tiles * sk1 / 512.0, tiles * sk2 / 1024.0
        splitk = sk1 if (f1 >= 0.9 or f2 <= f1 + 0.05) else sk2
        splitk = max(1, min(math.ceil(M / 64 / 4), splitk))
        bias_dst = bias_out if bias_out is not None else (grad_of(b_param) if b_param is not None else None)
        gw = grad_of(w_param)
        wg = wgrad_stream_for_current()
        if (_WGRAD_PENDING[0] is not None and dy.stride(1) == 1 and x.stride(1) == 1
                and hip.LIB.fn("mvuld_gemm_tn_wgrad_group_ok")(M, N, K, dy.stride(0), x.stride(0))):
            _WGRAD_PENDING[0].append((dy, x, gw, bias_dst, M, N, K))
            if bias_out is not None:          # the caller reads bias_out next: the group closes here
                return wgrad_group_flush()
            return wg if wg is not None else torch.cuda.current_stream(dy.device)
        hip.TIMING.annotate("gemm_tn_wgrad", 2.0 * M * N * K)
        if wg is None:
            _tn_wgrad_call(dy, x, gw, bias_dst, M, N, K, splitk)
        else:
            # Nothing downstream in backward reads a weight gradient: it leaves the critical path and runs on the weight-gradient
            # stream, ordered after the kernels that produced dy / x; the caching allocator is told both are still in use there.
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dy.device))
            wg.wait_event(ev)
            dy.record_stream(wg)
            x.record_stream(wg)
            if bias_out is not None:
                bias_out.record_stream(wg)
            with torch.cuda.stream(wg):
                _tn_wgrad_call(dy, x, gw, bias_dst, M, N, K, splitk)
        return wg if wg is not None else torch.cuda.current_stream(dy.device)
    if bias_out is not None:
        colsum_into(dy, bias_out)
    if b_param is not None:
        colsum_into(dy, grad_of(b_param))
    if w_param is None:
        return torch.cuda.current_stream(dy.device) if dy.is_cuda else None
    dyT = transpose(dy) if dyT is None else dyT
    xT = transpose(x) if xT is None else xT
    gw = grad_of(w_param)
    gemm_nt(dyT, xT, out=gw.view(N, K), out_mode=hip.OUT_ATOMIC, splitk=wgrad_splitk(N, K, M))
    return torch.cuda.current_stream(dy.device) if dy.is_cuda else None


_WS = {}


def _workspace(device, nbytes):
    """Persistent fp32 scratch per (device, stream): kernels that borrow it run in stream order, so reuse is ordered."""
    key = (device.type, device.index, torch.cuda.current_stream(device).cuda_stream)
    ws = _WS.get(key)
    if ws is None or ws.numel() * 4 < nbytes:
        ws = _WS[key] = torch.empty(nbytes // 4, dtype=torch.float32, device=device)
    return ws


def layernorm_fwd(x, gamma, beta, eps=1e-5, residual=None, rowscale=None, rows_per_sample=1, pre=None, want_sum=False, emit=None):
    """emit = Fp8Site: also leave y as e4m3 for the next fp8 product -> a fifth result, the (q, scale) pair."""
    rows, C = x.shape
    y = torch.empty_like(x)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    xsum = torch.empty_like(x) if (pre is not None and want_sum) else None
    if emit is False:                      # caller unpacks five results either way
        call("layernorm_fwd", ptr(x), ptr(pre), ptr(xsum), ptr(gamma), ptr(beta), ptr(residual), ptr(rowscale), rows_per_sample,
             ptr(y), ptr(mean), ptr(rstd), rows, C, eps, dt(x))
        return y, mean, rstd, xsum, None
    if emit is not None and emit.cal and x.dtype == torch.bfloat16 and C % 8 == 0:
        q = torch.empty((rows, C), dtype=torch.uint8, device=x.device)
        call("layernorm_fwd_q8", ptr(x), ptr(pre), ptr(xsum), ptr(gamma), ptr(beta), ptr(residual), ptr(rowscale), rows_per_sample,
             ptr(y), ptr(mean), ptr(rstd), rows, C, eps, ptr(q), ptr(emit.state))
        return y, mean, rstd, xsum, (q, emit.state[0:1])
    call("layernorm_fwd", ptr(x), ptr(pre), ptr(xsum), ptr(gamma), ptr(beta), ptr(residual), ptr(rowscale), rows_per_sample,
         ptr(y), ptr(mean), ptr(rstd), rows, C, eps, dt(x))
    if emit is not None:
        qpair = None
        if x.dtype == torch.bfloat16 and C % 8 == 0:            # first pass through this site: dynamic quantisation calibrates it
            q, _ = quant_fp8(y, scale=emit.state[0:1])
            emit.cal = True
            qpair = (q, emit.state[0:1])
        return y, mean, rstd, xsum, qpair
    return y, mean, rstd, xsum


USE_LN_DROP = [os.environ.get("MVULD_LN_DROP", "1") != "0"]


def layernorm_dropout_fwd(x, gamma, beta, eps, pre, p, seed, want_sum=True):
    """LayerNorm(dropout(x, p, seed) + pre): the hidden-state dropout of the text encoder's post-LN blocks in the LayerNorm's own pass
    (mvuld_layernorm_fwd_drop: same mask, same bits as dropout() followed by layernorm_fwd()).  -> (y, mean, rstd, xsum)."""
    rows, C = x.shape
    if not (USE_LN_DROP[0] and p > 0.0 and x.dtype == torch.bfloat16 and C % 8 == 0 and x.is_contiguous() and pre.is_contiguous()):
        y, mean, rstd, xsum = layernorm_fwd(dropout(x, p, seed), gamma, beta, eps, pre=pre, want_sum=want_sum)[:4]
        return y, mean, rstd, xsum
    y = torch.empty_like(x)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    xsum = torch.empty_like(x) if want_sum else None
    call("layernorm_fwd_drop", ptr(x), ptr(pre), ptr(xsum), ptr(gamma), ptr(beta), ptr(y), ptr(mean), ptr(rstd), rows, C, eps, float(p),
         int(seed) & 0xFFFFFFFFFFFFFFFF, rng_offset_ptr())
    return y, mean, rstd, xsum


def _ln_defer(x, rows, C):
    """(partials buffer, nparts) when this LayerNorm backward's parameter gradients can leave the chain: while a weight-gradient stream
    is active for the current stream (the fused step's image encoder: ops.fire_backward_done flushes); else None."""
    if not (USE_LN_DEFER[0] and wgrad_stream_for_current() is not None):
        return None
    nbytes = hip.LIB.fn("mvuld_layernorm_bwd_workspace_bytes")(C)
    nparts = hip.LIB.fn("mvuld_layernorm_bwd_nparts")(rows, C, nbytes, dt(x))
    if nparts <= 0:
        return None
    return torch.empty(max(nparts, 64) * 2 * C, dtype=torch.float32, device=x.device), nparts      # (>= 64 rows: the size that selects the partial form)


def layernorm_bwd(dy, x, gamma_p, beta_p, mean, rstd, rowscale=None, rows_per_sample=1):
    rows, C = x.shape
    dx = torch.empty_like(x)
    d = _ln_defer(x, rows, C)
    if d is not None:
        part, nparts = d
        call("layernorm_bwd", ptr(dy), ptr(x), ptr(gamma_p), ptr(mean), ptr(rstd), ptr(rowscale), rows_per_sample, ptr(dx),
             None, None, rows, C, ptr(part), part.numel() * 4, dt(x))
        _LN_PENDING.append((part, nparts, C, grad_of(gamma_p), grad_of(beta_p), torch.cuda.current_stream(x.device)))
        return dx
    ws = _workspace(x.device, hip.LIB.fn("mvuld_layernorm_bwd_workspace_bytes")(C))
    call("layernorm_bwd", ptr(dy), ptr(x), ptr(gamma_p), ptr(mean), ptr(rstd), ptr(rowscale), rows_per_sample, ptr(dx),
         ptr(grad_of(gamma_p)), ptr(grad_of(beta_p)), rows, C, ptr(ws), ws.numel() * 4, dt(x))
    return dx


def layernorm_bwd_dropout(dy, x, gamma_p, beta_p, mean, rstd, p, seed):
    """-> (dx, dropout(dx, p, seed)): layernorm_bwd with the backward hidden dropout written by the same pass (mvuld_layernorm_bwd_drop:
    same mask, same bits as the two launches)."""
    rows, C = x.shape
    if not (USE_LN_DROP[0] and p > 0.0 and x.dtype == torch.bfloat16 and C % 8 == 0 and x.is_contiguous() and dy.is_contiguous()):
        dx = layernorm_bwd(dy, x, gamma_p, beta_p, mean, rstd)
        return dx, dropout(dx, p, seed)
    dx, dxd = torch.empty_like(x), torch.empty_like(x)
    ws = _workspace(x.device, hip.LIB.fn("mvuld_layernorm_bwd_workspace_bytes")(C))
    call("layernorm_bwd_drop", ptr(dy), ptr(x), ptr(gamma_p), ptr(mean), ptr(rstd), ptr(dx), ptr(dxd), ptr(grad_of(gamma_p)), ptr(grad_of(beta_p)),
         rows, C, ptr(ws), ws.numel() * 4, float(p), int(seed) & 0xFFFFFFFFFFFFFFFF, rng_offset_ptr())
    return dx, dxd


def batchnorm_fwd(x, gamma, beta, run_mean, run_var, O, C, I, so, sc, si, training, eps=1e-5, momentum=0.1):
    y = torch.empty_like(x)
    sm = torch.empty(C, dtype=torch.float32, device=x.device)
    sr = torch.empty(C, dtype=torch.float32, device=x.device)
    call("batchnorm_fwd", ptr(x), ptr(y), ptr(gamma), ptr(beta), ptr(run_mean), ptr(run_var), ptr(sm), ptr(sr), O, C, I, so, sc, si,
         eps, momentum, 1 if training else 0, dt(x))
    return y, sm, sr


def batchnorm_bwd(dy, x, gamma_p, beta_p, sm, sr, O, C, I, so, sc, si, training, dxsum=None):
    """dxsum: optional fp32 [C] buffer that receives += the per-channel sum of dx (before rounding to x.dtype)."""
    dx = torch.empty_like(x)
    call("batchnorm_bwd", ptr(dy), ptr(x), ptr(gamma_p), ptr(sm), ptr(sr), ptr(dx), ptr(grad_of(gamma_p)), ptr(grad_of(beta_p)),
         O, C, I, so, sc, si, 1 if training else 0, ptr(dxsum), dt(x))
    return dx


def act_bwd(dy, ref, mode):
    dx = torch.empty_like(dy)
    call("act_bwd", ptr(dy), ptr(ref), ptr(dx), dy.numel(), mode, dt(dy))
    return dx


def cast(x, dtype):
    if x.dtype == dtype:
        return x
    y = torch.empty(x.shape, dtype=dtype, device=x.device)
    call("cast", ptr(x), dt(x), ptr(y), hip.F32 if dtype == torch.float32 else hip.BF16, x.numel())
    return y


def add(a, b):
    y = torch.empty_like(a)
    call("add", ptr(a), ptr(b), ptr(y), a.numel(), dt(a))
    return y


def mul(a, b):
    y = torch.empty_like(a)
    call("mul", ptr(a), ptr(b), ptr(y), a.numel(), dt(a))
    return y


# Device-resident step counter (int64 [1]) mixed into every RNG kernel's seed, or None.  A hipGraph-captured training step sets it
# (graph_step.GraphedTrainStep): host seeds are frozen into the captured launches, the counter advances inside the graph.
RNG_OFFSET = [None]


def seed_rng(seed, rank=0):
    """Start every dropout / DropPath mask stream of the package from (seed, rank), as the reference seeds torch with `seed + rank`
    (main_bigvul.py:536-540): data-parallel ranks then draw DIFFERENT masks for the same sample slot, and runs with different --seed
    differ.  Without a call the streams start from fixed constants (tests rely on that)."""
    import hashlib
    from .models import GraphModel, unixcoder

    def mix(tag):
        h = hashlib.blake2b(f"{tag}/{int(seed)}/{int(rank)}".encode(), digest_size=8).digest()
        return int.from_bytes(h, "little") | 1
    GraphModel._SEED[0] = mix("head")
    unixcoder._SEED[0] = mix("text")
    SWIN_DROPPATH_SEED[0] = mix("swin")


SWIN_DROPPATH_SEED = [0x0D50F7A7]          # initial DropPath stream of every SwinTransformerV2 built after this (seed_rng resets it)


def rng_offset_ptr():
    return ptr(RNG_OFFSET[0])


def dropout(x, p, seed):
    if p <= 0.0:
        return x
    y = torch.empty_like(x)
    call("dropout", ptr(x), ptr(y), x.numel(), float(p), int(seed) & 0xFFFFFFFFFFFFFFFF, rng_offset_ptr(), dt(x))
    return y


# ---------------------------------------------------------------------------------------------
class AttnGeom:
    """Geometry of one fused-attention call (see include/mvuld_hip.h)."""

    def __init__(self, mode, B, H, hd, N, nW=1, res=0, ws=0, shift=0, scale=1.0, sumsq=0, drop_p=0.0, drop_seed=0):
        self.mode, self.B, self.H, self.hd, self.N, self.nW = mode, B, H, hd, N, nW
        self.res, self.ws, self.shift, self.scale = res, ws, shift, scale
        self.drop_p, self.drop_seed = float(drop_p), int(drop_seed) & 0xFFFFFFFFFFFFFFFF     # attention-probability dropout (modes 1, 2)
        self.sumsq = sumsq              # mode 2: sum of squared sequence lengths (algorithmic FLOP count of the instrumented step)

    def args(self):
        return (self.mode, self.B, self.H, self.hd, self.N, self.nW, self.res, self.ws, self.shift, float(self.scale))


def _mfma_attn_ok(g: AttnGeom, dtype):
    """The matrix-core kernels keep K and V (or Q~ and dO) of one (window, head) resident in LDS: <= 160 KiB."""
    if ATTN_IMPL[0] != "auto" or dtype != torch.bfloat16:
        return False
    npad = (g.N + 31) // 32 * 32
    t2 = (2 * g.ws - 1) ** 2 if g.mode == 0 else 0
    need = max(2 * npad * (g.hd + 8) * 2 + npad * 4 + 32 + 2 * t2 * 4, 2 * npad * (g.hd + 8) * 2 + npad * 12 + t2 * 4)
    if g.mode == 0 and (g.hd != 32 or g.ws > 32):
        return False
    if g.mode == 2 and need > 160 * 1024:
        raise RuntimeError(f"packed attention: sequences of up to {g.N} tokens do not fit the matrix-core kernel's LDS")
    return need <= 160 * 1024


def attn_fwd(g: AttnGeom, qkv, table16=None, logit_scale=None, valid=None, sample_scale=None):
    tokens = qkv.shape[0]
    out = torch.empty((tokens, g.H * g.hd), dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty((g.B * g.nW * g.H * g.N,), dtype=torch.float32, device=qkv.device)
    name = "attn_fwd_mfma" if _mfma_attn_ok(g, qkv.dtype) else "attn_fwd_simple"
    if g.mode == 2 and name != "attn_fwd_mfma":
        raise RuntimeError("packed (mode 2) attention exists on the matrix-core path only (bf16)")
    hip.TIMING.annotate(name, 4.0 * (g.sumsq if g.mode == 2 else g.N * g.N * g.B * g.nW) * g.hd * g.H)
    if name == "attn_fwd_mfma":
        call(name, *g.args(), ptr(qkv), ptr(table16), ptr(logit_scale), ptr(valid), ptr(out), ptr(lse), g.drop_p, g.drop_seed,
             rng_offset_ptr(), ptr(sample_scale) if USE_DROPPATH_SKIP[0] else None, dt(qkv))
    else:
        if g.drop_p > 0.0:
            raise RuntimeError("attention-probability dropout exists on the matrix-core attention path only (bf16)")
        call(name, *g.args(), ptr(qkv), ptr(table16), ptr(logit_scale), ptr(valid), ptr(out), ptr(lse), dt(qkv))
    return out, lse


# DropPath as a saving: the attention kernels write zeros for the samples whose residual branch is dropped in this block (sample_scale == 0)
USE_DROPPATH_SKIP = [os.environ.get("MVULD_DROPPATH_SKIP", "1") != "0"]
BIAS_STREAM = [None]       # stream on which the last attn_bwd left dtable16 (None = the current one)


def attn_bwd_is_fused(g: AttnGeom) -> bool:
    """True when mvuld_attn_bwd_mfma takes this geometry with the fused single-pass kernel (dQ, dK, dV and the table gradient on the
    caller's stream, in one call)."""
    return g.mode == 0 and hip.LIB.fn("mvuld_attn_bwd_fused_active")(0, g.hd, g.ws) == 1


def attn_bwd(g: AttnGeom, qkv, out, dout, lse, table16=None, logit_scale=None, valid=None, dtable16=None, dlogit_scale=None, sample_scale=None):
    dqkv = torch.empty_like(qkv)
    ss = ptr(sample_scale) if USE_DROPPATH_SKIP[0] else None
    BIAS_STREAM[0] = None
    if _mfma_attn_ok(g, qkv.dtype):
        delta = torch.empty((qkv.shape[0] * g.H,), dtype=torch.float32, device=qkv.device)
        qt = torch.empty((qkv.shape[0], g.H * g.hd), dtype=torch.bfloat16, device=qkv.device) if g.mode == 0 else None
        hip.TIMING.annotate("attn_bwd_mfma", 10.0 * (g.sumsq if g.mode == 2 else g.N * g.N * g.B * g.nW) * g.hd * g.H)      # algorithmic: five products
        args = (*g.args(), ptr(qkv), ptr(table16), ptr(logit_scale), ptr(valid), ptr(out), ptr(dout), ptr(lse),
                ptr(dqkv), ptr(dtable16), ptr(dlogit_scale), ptr(delta), ptr(qt))
        wg = wgrad_stream_for_current() if g.mode == 0 else None
        # round 4: the fused window backward forms the table gradient in the same pass as dQ / dK / dV -- one call, this stream
        fused = attn_bwd_is_fused(g)
        if g.mode != 0 or wg is None or fused:
            part = _workspace(qkv.device, hip.LIB.fn("mvuld_attn_bwd_mfma_workspace_bytes")(0, g.B, g.H, g.nW, g.ws)) if g.mode == 0 else None
            call("attn_bwd_mfma", *args, ptr(part), part.numel() * 4 if part is not None else 0, 3, g.drop_p, g.drop_seed,
                 rng_offset_ptr(), ss, dt(qkv))
            return dqkv
        # The bias-table gradient feeds nothing else in backward: dQ / dK / dV stay on this stream, the table pass (and whatever
        # the caller does with dtable16 afterwards: see bias_stream) goes to the weight-gradient stream.
        call("attn_bwd_mfma", *args, None, 0, 1, 0.0, 0, None, ss, dt(qkv))
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(qkv.device))
        wg.wait_event(ev)
        for t in (qkv, dout, lse, delta, qt, table16, logit_scale, dtable16, out):
            t.record_stream(wg)
        with torch.cuda.stream(wg):
            part = _workspace(qkv.device, hip.LIB.fn("mvuld_attn_bwd_mfma_workspace_bytes")(0, g.B, g.H, g.nW, g.ws))
            call("attn_bwd_mfma", *args, ptr(part), part.numel() * 4, 2, 0.0, 0, None, ss, dt(qkv))
        BIAS_STREAM[0] = wg
        return dqkv
    hip.TIMING.annotate("attn_bwd_simple", 10.0 * g.N * g.N * g.hd * g.H * g.B * g.nW)
    call("attn_bwd_simple", *g.args(), ptr(qkv), ptr(table16), ptr(logit_scale), ptr(valid), ptr(out), ptr(dout), ptr(lse),
         ptr(dqkv), ptr(dtable16), ptr(dlogit_scale), dt(qkv))
    return dqkv
