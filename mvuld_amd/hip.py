"""ctypes binding of ``libmvuld_hip.so`` (C ABI declared in ``include/mvuld_hip.h``).

Prototypes are parsed from the header itself, so the binding cannot drift from the
declared ABI.  There is NO fallback: if the library is missing or a call fails, the
product path raises -- nothing here routes to PyTorch compute or to ``oracle/``.
"""
import ctypes
import os
import re

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MVULD_HIP_LIB") or os.path.join(_HERE, "libmvuld_hip.so")     # override: A/B runs of a variant build
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "mvuld_hip.h")

F32, BF16 = 0, 1
EPI_NONE, EPI_BIAS, EPI_GELU, EPI_ELU, EPI_MUL_DGELU, EPI_MUL_DELU, EPI_ADD_AUX, EPI_GELU_DG, EPI_MUL_AUX = range(9)
OUT_STORE, OUT_ACCUM, OUT_ATOMIC = range(3)

_CT = {
    "int": ctypes.c_int, "float": ctypes.c_float, "int64_t": ctypes.c_int64, "uint64_t": ctypes.c_uint64,
    "mvuld_stream_t": ctypes.c_void_p,
}


def parse_header(path=HEADER_PATH):
    """{name: (restype, [argtypes])} for every ``mvuld_*`` prototype in the header."""
    txt = open(path).read()
    txt = re.sub(r"/\*.*?\*/", " ", txt, flags=re.S)
    protos = {}
    for m in re.finditer(r"(const\s+char\s*\*|int64_t|int)\s+(mvuld_\w+)\s*\(([^;{]*?)\)\s*;", txt, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        argtypes = []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                if "*" in a:
                    argtypes.append(ctypes.c_void_p)
                else:
                    base = a.replace("const ", "").split(" ")[0]
                    argtypes.append(_CT[base])
        protos[name] = (ctypes.c_char_p if "char" in ret else (ctypes.c_int64 if ret == "int64_t" else ctypes.c_int), argtypes)
    return protos


class _Lib:
    def __init__(self):
        self._dll = None
        self._protos = None

    def load(self):
        if self._dll is None:
            if not os.path.exists(LIB_PATH):
                raise RuntimeError(
                    f"mvuld_amd: {LIB_PATH} is missing -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                    "(make -C mvuld_amd/csrc).  There is no CPU / PyTorch fallback for the hot path.")
            self._dll = ctypes.CDLL(LIB_PATH)
            self._protos = parse_header()
            for name, (ret, argtypes) in self._protos.items():
                fn = getattr(self._dll, name)         # AttributeError => header/library drift
                fn.restype = ret
                fn.argtypes = argtypes
        return self._dll

    def fn(self, name):
        return getattr(self.load(), name)


LIB = _Lib()


def available() -> bool:
    return os.path.exists(LIB_PATH)


def last_error() -> str:
    return LIB.fn("mvuld_last_error")().decode()


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    return t.data_ptr()


def dt(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise TypeError(f"mvuld_amd: unsupported activation dtype {t.dtype}")


def stream():
    return torch.cuda.current_stream().cuda_stream


class _Timing:
    """Optional per-launch timing with HIP events on the launch stream (bench.py's instrumented step)."""

    def __init__(self):
        self.enabled = False
        self.records = []
        self.pending = None
        self.bracket_us = None     # per-launch cost of the event pair itself (calibrate()), subtracted in summary()

    def enable(self):
        self.enabled, self.records, self.pending = True, [], None
        if self.bracket_us is None:
            self.calibrate()

    def calibrate(self):
        """What an event pair adds to the launch it brackets: the time between the first event's timestamp and the kernel's start, and
        between its end and the second timestamp (5-8 us on this stack -- 10-15 % of a 50 us GEMM, which is why the instrumented step's
        family sums sat above the rocprofv3 durations of the same launches).  Measured as: median elapsed time of a bracketed tiny launch
        (a 256-element cast) minus that launch's back-to-back cost (one bracket around 256 of them, an UPPER bound on its duration): a
        LOWER bound on the overhead, so the corrected durations stay upper bounds on the kernels' own time."""
        self.bracket_us = 0.0
        if not torch.cuda.is_available():
            return
        x = torch.zeros(256, dtype=torch.float32, device="cuda")
        y = torch.empty(256, dtype=torch.bfloat16, device="cuda")
        fn = LIB.fn("mvuld_cast")

        def launch():
            fn(x.data_ptr(), F32, y.data_ptr(), BF16, 256, stream())
        for _ in range(16):
            launch()
        torch.cuda.synchronize()
        pairs = []
        for _ in range(64):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            launch()
            e1.record()
            pairs.append((e0, e1))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(256):
            launch()
        e1.record()
        torch.cuda.synchronize()
        single = sorted(a.elapsed_time(b) for a, b in pairs)[len(pairs) // 2] * 1e3
        chained = e0.elapsed_time(e1) * 1e3 / 256
        self.bracket_us = max(0.0, single - chained)

    def disable(self):
        self.enabled = False

    def annotate(self, family, flops=0.0, nbytes=0.0):
        """Name the kernel family and the ALGORITHMIC work of the next call."""
        if self.enabled:
            self.pending = (family, float(flops), float(nbytes))

    def summary(self):
        torch.cuda.synchronize()
        fam = {}
        over = (self.bracket_us or 0.0) * 1e-3
        for name, fl, by, e0, e1 in self.records:
            d = fam.setdefault(name, {"ms": 0.0, "n": 0, "flops": 0.0, "bytes": 0.0})
            t = e0.elapsed_time(e1)
            d["ms"] += max(0.25 * t, t - over)         # (never more than three quarters of a measurement: tiny launches keep a floor)
            d["n"] += 1
            d["flops"] += fl
            d["bytes"] += by
        return fam


TIMING = _Timing()


def call(name, *args):
    """Invoke ``mvuld_<name>`` on the current stream; raise on a non-zero status."""
    fn = LIB.fn("mvuld_" + name)
    if TIMING.enabled:
        fam, fl, by = TIMING.pending or (name, 0.0, 0.0)
        TIMING.pending = None
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = fn(*args, stream())
        e1.record()
        TIMING.records.append((fam, fl, by, e0, e1))
    else:
        rc = fn(*args, stream())
    if rc != 0:
        raise RuntimeError(f"mvuld_{name} failed ({rc}): {last_error()}")


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("mvuld_amd: the hot path runs only on the GPU through libmvuld_hip.so "
                               "(got a CPU tensor; there is no CPU fallback)")
