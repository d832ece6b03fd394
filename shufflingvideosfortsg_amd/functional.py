"""torch.autograd.Function wrappers over the C ABI (one per fused kernel pair).

All inputs must be CUDA(HIP) fp32 tensors; they are made contiguous here; outputs and workspaces
are allocated from torch's caching allocator and handed to the library as raw pointers; the
kernels run on torch's current stream.
"""
from __future__ import annotations

import ctypes
import os
import torch

from . import _lib
from ._lib import TSG_BF16, TSG_F32, TSG_F32S, check, load, ptr, require_device, stream_of


class _KernelTimer:
    """Optional per-launch timing of the C-ABI calls: an event pair on the launch stream around each
    call (bench.py reads the means).  Disabled by default: no events, no overhead."""

    def __init__(self):
        self.on = False
        self.records = []
        self.slots = []
        self.only = None

    def enable(self, only=None):
        """``only``: tuple of entry-point name prefixes to time (None = every call).  An event pair is a pair of barrier
        packets on the queue: around every one of the ~45 short operand-split launches of an "f32s" step they cost 3.5 ms,
        so bench.py times only the hot-path kernels its roofline needs."""
        self.on, self.records, self.only, self.slots = True, [], only, []

    def disable(self):
        self.on = False

    def summary(self):
        """(name, dims) -> (mean microseconds, launches, median microseconds); dims = the integer shape arguments of
        the call.  Synchronises on the recorded events."""
        acc = {}
        for name, e0, e1, dims in self.records:
            e1.synchronize()
            acc.setdefault((name, dims), []).append(e0.elapsed_time(e1) * 1e3)
        return {k: (sum(v) / len(v), len(v), sorted(v)[len(v) // 2]) for k, v in acc.items()}

    def kernel_summary(self):
        """The same keys for the launches that were ALSO bracketed by their own event pair (tsg_time_next_launch: the K1 / K1g
        forward and fused backward kernels): the kernel's duration alone, as rocprofv3's kernel trace reports it -- the pair
        recorded around the call (``summary``) additionally holds the dispatch gap in front of the kernel."""
        acc = {}
        us = ctypes.c_float()
        for name, dims, slot in self.slots:
            if load().tsg_timed_launch_us(slot, ctypes.byref(us)) == 0:
                acc.setdefault((name, dims), []).append(float(us.value))
        return {k: (sum(v) / len(v), len(v), sorted(v)[len(v) // 2]) for k, v in acc.items()}


kernel_timer = _KernelTimer()
_SELF_TIMED = ("tsg_scdm_attn_fwd", "tsg_scdm_gate_fwd", "tsg_scdm_attn_bwd", "tsg_scdm_gate_bwd")   # entry points that honour tsg_time_next_launch


class KernelWaitExpired(RuntimeError):
    """A bounded wait inside a kernel expired (persistent LSTM hand-off, K1 backward partner exchange): the outputs of that
    launch are invalid."""


LstmWaitExpired = KernelWaitExpired     # the name of rounds 1-2 (the LSTM was the only kernel with a bounded wait then)

_err_sink = None           # pinned host word every kernel with a bounded wait sets on expiry (tsg_error_sink)
_err_words = {}            # device index -> int32 [1] device tensor set on expiry as well (tsg_error_word): the optimizer guard reads it
_selftest_done = False


def _register_error_channels(device) -> None:
    """Register the process-wide expiry channels with the library: the pinned host word (polled by ``check_kernel_errors``) and
    the current device's error word (read on the device by ``engine.optimizer_step``).  Called from ``_call`` on the first launch
    of ANY entry point (the K1 backward's exchange reported into an unregistered sink when no LSTM call had run)."""
    global _err_sink
    lib = load()
    if _err_sink is None:
        _err_sink = torch.zeros(1, dtype=torch.int32).pin_memory()
        check(lib.tsg_error_sink(_err_sink.data_ptr()), "tsg_error_sink")
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if idx not in _err_words:
        _err_words[idx] = torch.zeros(1, dtype=torch.int32, device=torch.device("cuda", idx))
        # the library keeps one word per device, registered for the CURRENT device of the call (ABI revision 5)
        with torch.cuda.device(idx):
            check(lib.tsg_error_word(_err_words[idx].data_ptr()), "tsg_error_word")


def error_word(device=None) -> torch.Tensor:
    """The device-resident expiry word (int32 [1]; non-zero after a launch whose bounded wait expired)."""
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    _register_error_channels(device)
    return _err_words[device.index if device.index is not None else torch.cuda.current_device()]


def check_kernel_errors() -> None:
    """Raise if a launch reported an expired bounded wait since the last check (its outputs are invalid).  Reads a pinned host
    word -- no synchronisation; called on every LSTM launch, by GraphedTrainStep on every replay, and by bench.py / the tests at
    the end of a region."""
    if _err_sink is not None and int(_err_sink[0]) != 0:
        _err_sink[0] = 0
        for wd in _err_words.values():
            wd.zero_()
        _k3_ws.clear()       # a launch that gave up may have left the cached K3 workspaces' tickets non-zero: the next call gets fresh, zeroed ones
        raise KernelWaitExpired("a kernel's bounded wait on other workgroups expired (persistent LSTM hand-off or the K1 backward's "
                                "partner exchange: workgroups not co-resident, e.g. the GPU is shared with another process?): the "
                                "outputs of that launch are invalid and the optimizer update of that step was skipped.  "
                                "TSG_LSTM_PERSIST=0 selects the launch-per-step LSTM kernels, TSG_K1_BWD=split the two-kernel K1 backward.")


_persistent_gate = None      # dp.FlatGradAllReduce (gated mode): fences its collectives against the persistent / partner-exchange kernels


def set_persistent_gate(gate) -> None:
    """Register the object whose ``before_persistent()`` / ``after_persistent()`` are called around every backward launch that needs
    the whole chip to itself (the persistent LSTM kernels: one workgroup per CU, all co-resident; the K1 backward's partner exchange:
    ``before`` only).  ``None`` removes it.  See ``dp.FlatGradAllReduce`` (gated=True)."""
    global _persistent_gate
    _persistent_gate = gate


def _gate_before():
    if _persistent_gate is not None:
        _persistent_gate.before_persistent()


def _gate_after():
    if _persistent_gate is not None:
        _persistent_gate.after_persistent()


def check_lstm_errors() -> None:
    """``check_kernel_errors`` preceded, once per process, by the persistent-LSTM self-test (before the first LSTM launch)."""
    global _selftest_done
    if not _selftest_done:
        _selftest_done = True
        _register_error_channels(torch.device("cuda", torch.cuda.current_device()))
        _lstm_selftest()
    check_kernel_errors()


def _lstm_selftest() -> None:
    """Once per process, before the first LSTM launch: a short persistent forward + backward at the full-chip grid
    ([128, 8, 512], 256 workgroups) on scratch data, synchronised.  If a bounded wait expires with the L2-local exchange
    allowed, the exchange falls back to write-through stores for the rest of the process (and the test is repeated); if it
    expires again the persistent kernels cannot run here (the GPU is shared?) and the process switches to the launch-per-step
    kernels (tsg_lstm_set_persist(0), a warning): slower, but correct."""
    import warnings
    lib = load()
    dev = torch.device("cuda", torch.cuda.current_device())
    B, T, h = 128, 8, 512
    st = torch.cuda.current_stream(dev).cuda_stream
    Gx = torch.zeros(T, B, 2, 4 * h, device=dev); W = torch.zeros(2, 4 * h, h, device=dev)
    out = torch.empty(T, B, 2 * h, device=dev); R = torch.empty(T, 2, B, h, 4, device=dev); Cs = torch.empty(T, 2, B, h, device=dev)
    dG = torch.empty(T, B, 2, 4 * h, device=dev); dC = torch.empty(2, B, h, device=dev)
    nb = int(lib.tsg_lstm_bwd_ws_bytes(B, T, h))
    ws = torch.empty(nb // 4 + 4, device=dev)
    for attempt in range(2):
        sync = torch.empty(512, device=dev, dtype=torch.int32)
        check(lib.tsg_lstm_fwd_bias(ptr(Gx), None, ptr(W), ptr(out), ptr(R), ptr(Cs), ptr(sync), B, T, h, TSG_F32S, 0, st), "tsg_lstm_fwd_bias")
        check(lib.tsg_lstm_bwd_ws_layout(ptr(W), ptr(R), ptr(Cs), ptr(out), None, ptr(dG), ptr(dC), ptr(ws), nb, None, B, T, h,
                                         TSG_F32S, 0, st), "tsg_lstm_bwd_ws_layout")
        torch.cuda.synchronize(dev)
        if int(_err_sink[0]) == 0:
            return
        _err_sink[0] = 0
        for wd in _err_words.values():
            wd.zero_()
        if attempt == 0:
            warnings.warn("persistent LSTM self-test: a bounded wait expired with the L2-local exchange; falling back to "
                          "write-through exchange stores for this process (TSG_LSTM_L2X=0)")
            check(lib.tsg_lstm_set_l2_exchange(0), "tsg_lstm_set_l2_exchange")
    warnings.warn("persistent LSTM self-test failed with both exchange modes: the persistent kernels cannot run on this device "
                  "(is the GPU shared?); using the launch-per-step kernels for this process (TSG_LSTM_PERSIST=0)")
    check(lib.tsg_lstm_set_persist(0), "tsg_lstm_set_persist")


def _call(name: str, like: torch.Tensor, *args) -> None:
    """Invoke one C-ABI entry point on ``like``'s current stream and raise on a non-zero return."""
    fn = getattr(load(), name)
    st = stream_of(like)
    if like.is_cuda and like.device.index not in _err_words:
        _register_error_channels(like.device)
    if kernel_timer.on and (kernel_timer.only is None or name.startswith(kernel_timer.only)):
        stream = torch.cuda.current_stream(like.device)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        dims = tuple(a for a in args if isinstance(a, int) and not isinstance(a, bool) and 0 <= a < (1 << 24))
        if name.startswith(_SELF_TIMED) and len(kernel_timer.slots) < 1024 and load().tsg_time_next_launch(len(kernel_timer.slots)) == 0:
            kernel_timer.slots.append((name, dims, len(kernel_timer.slots)))
        e0.record(stream)
        rc = fn(*args, st)
        e1.record(stream)
        load().tsg_time_next_launch(-1)                     # (disarm: an entry point that took another kernel path leaves the slot unused)
        kernel_timer.records.append((name, e0, e1, dims))
    else:
        rc = fn(*args, st)
    check(rc, name)


# Library-GEMM precision of the adjacent glue (the LSTM's input / weight-gradient GEMMs): None = fp32 rocBLAS
# (parity mode, default); torch.bfloat16 = bf16 operands with fp32 accumulation (BASELINE configs 2-4);
# "f32s" = split precision: both operands as bf16 (hi, lo) planes (tsg_split_bf16x3), ONE bf16 MFMA GEMM over the
# 3x longer contraction with fp32 accumulate = hi·hi + hi·lo + lo·hi, error <= 2e-5 x scale against float64 (2^-16 products: ~100x a true fp32 GEMM's rounding) at 2-3x its speed.
# "bf16" (the string) = bf16 STORAGE mode (BASELINE configs 2 / 4): every activation and activation
# gradient of the path lives in HBM as bf16 -- the hand-written kernels take dtype TSG_BF16 (half the bytes; fp32
# arithmetic, softmax, cell state and accumulation inside), the GEMMs are plain bf16 MFMA GEMMs with fp32 accumulation
# (no 3x contraction, no operand passes), parameters, weight gradients and the optimizer stay fp32.
# The hand-written kernels always compute in fp32.
_GEMM_DTYPE = None


def set_gemm_dtype(dtype=None):
    global _GEMM_DTYPE
    if dtype not in (None, torch.float32, torch.bfloat16, "f32s", "bf16"):
        raise ValueError("gemm dtype must be None/float32, bfloat16 (library-GEMM operands only), 'f32s' or 'bf16' (bf16 storage)")
    _GEMM_DTYPE = None if dtype in (None, torch.float32) else dtype


def get_gemm_dtype():
    """The current mode as ``set_gemm_dtype`` takes it (None = strict fp32)."""
    return _GEMM_DTYPE


def bf16_storage() -> bool:
    """True in the bf16 storage mode (``engine.precision("bf16")``)."""
    return isinstance(_GEMM_DTYPE, str) and _GEMM_DTYPE == "bf16"


def split_bf16x3(x: torch.Tensor, k_dim: int, right: bool) -> torch.Tensor:
    """fp32 contiguous [R,C] -> bf16 with the contraction dimension ``k_dim`` tripled: planes (hi,hi,lo) for a
    left GEMM operand, (hi,lo,hi) for a right one (include/tsg_hip.h: tsg_split_bf16x3)."""
    require_device(x)
    x = _f32c(x)
    R, C = x.shape
    if k_dim == 1:
        out = torch.empty(R, 3 * C, device=x.device, dtype=torch.bfloat16)
        ld, plane = 3 * C, C
    else:
        out = torch.empty(3 * R, C, device=x.device, dtype=torch.bfloat16)
        ld, plane = C, R * C
    _call("tsg_split_bf16x3", x, ptr(x), ptr(out), R, C, ld, plane, int(right))
    return out


def split_bf16x3_rows_shifted(x: torch.Tensor, col0: int, cols: int, row_shift: int, right: bool,
                              out: torch.Tensor = None, out_col0: int = 0, period: int = 0) -> torch.Tensor:
    """Columns [col0, col0+cols) of fp32 contiguous x [R,C], rows shifted down by ``row_shift`` (zeros shifted in), as
    the row-stacked bf16 planes [3R, cols] of a GEMM operand contracted over the rows (tsg_split_bf16x3_shift) -- the
    h_{t-1} operand of the LSTM weight-gradient GEMM without the shifted copy.  With ``out`` [3R, W] given, the planes
    go into its columns [out_col0, out_col0+cols) (several operands side by side for one GEMM)."""
    require_device(x)
    x = _f32c(x)
    R, C = x.shape
    if out is None:
        out = torch.empty(3 * R, cols, device=x.device, dtype=torch.bfloat16)
    W = out.shape[1]
    if out.shape[0] != 3 * R or not out.is_contiguous() or out.dtype != torch.bfloat16 or out_col0 + cols > W or (out_col0 | W) % 4:
        raise ValueError("split_bf16x3_rows_shifted: bad output buffer")
    _call("tsg_split_bf16x3_shift", x, x.data_ptr() + 4 * col0, C, row_shift, period, out.data_ptr() + 2 * out_col0, R, cols, W, R * W, int(right))
    return out


def split_bf16x3_t(x: torch.Tensor, col0: int, cols: int, row_shift: int, right: bool, out: torch.Tensor, out_row0: int = 0,
                   period: int = 0, dup_row0: int = None):
    """Transposing split (tsg_split_bf16x3_t): columns [col0, col0+cols) of fp32 contiguous x [R,C], rows shifted by
    ``row_shift`` (within sequences of ``period`` rows when period > 0) -> rows [out_row0, out_row0+cols) of the bf16 buffer ``out`` [W, 3R] whose row c holds the three planes
    of input column c one after the other (contraction index contiguous)."""
    require_device(x)
    x = _f32c(x)
    R, C = x.shape
    if out.dim() != 2 or out.shape[1] != 3 * R or not out.is_contiguous() or out.dtype != torch.bfloat16 or out_row0 + cols > out.shape[0]:
        raise ValueError("split_bf16x3_t: bad output buffer")
    if dup_row0 is not None and (dup_row0 + cols > out.shape[0] or abs(dup_row0 - out_row0) < cols):
        raise ValueError("split_bf16x3_t: bad duplicate destination")
    dup = 0 if dup_row0 is None else (dup_row0 - out_row0) * 3 * R                 # second copy at rows [dup_row0, dup_row0+cols)
    _call("tsg_split_bf16x3_t", x, x.data_ptr() + 4 * col0, C, row_shift, period, out.data_ptr() + 2 * out_row0 * 3 * R, R, cols, 3 * R, R,
          int(right), dup)
    return out


def transposed(w: torch.Tensor) -> torch.Tensor:
    """``w.transpose(-1, -2).contiguous()`` of an fp32 matrix (a column slice is read in place) or a contiguous batch of them on
    tsg_transpose_f32 -- the weight operand of the input-gradient GEMMs and W_hh^T of the LSTM backward; torch's strided copy elsewhere."""
    if (_OWN_TRANSPOSE and w.is_cuda and w.dtype == torch.float32 and w.dim() in (2, 3) and w.shape[-1] % 4 == 0 and w.shape[-2] % 4 == 0
            and w.stride(-1) == 1 and w.stride(-2) % 4 == 0 and w.stride(-2) >= w.shape[-1] and w.data_ptr() % 16 == 0
            and (w.dim() == 2 or w.shape[0] == 1 or w.stride(0) == w.shape[1] * w.stride(1))):
        out = torch.empty(*w.shape[:-2], w.shape[-1], w.shape[-2], device=w.device, dtype=w.dtype)
        _call("tsg_transpose_f32", w, ptr(w), w.stride(-2), ptr(out), w.shape[0] if w.dim() == 3 else 1, w.shape[-2], w.shape[-1])
        return out
    return w.transpose(-1, -2).contiguous()


def wgrad_f32s_ok(M: int, N: int, K0: int, K1: int = 0) -> bool:
    """Shapes tsg_wgrad_f32s takes (include/tsg_hip.h)."""
    return M > 0 and M % 32 == 0 and N > 0 and N % 256 == 0 and K0 % 128 == 0 and K1 % 128 == 0 and K0 + K1 > 0


def _rows2d(t: torch.Tensor) -> torch.Tensor:
    """fp32 2-D operand whose rows are contiguous (a column slice of a row-major matrix is taken as it is)."""
    if t.dtype != torch.float32 or t.dim() != 2:
        raise TypeError(f"fp32 matrix expected, got {t.dtype} {tuple(t.shape)}")
    return t if t.stride(1) == 1 and t.stride(0) % 4 == 0 and t.data_ptr() % 16 == 0 else t.contiguous()


def wgrad_f32s(A: torch.Tensor, B0: torch.Tensor, N: int = None, groups: int = 1, a_group_stride: int = 0,
               B1: torch.Tensor = None, K1: int = 0, b1_group_stride: int = 0, shift: int = 0, period: int = 0,
               out: torch.Tensor = None) -> torch.Tensor:
    """C[g] = A[:, g*a_group_stride : +N]^T @ [B0 | B1[rows shifted by -+shift, g*b1_group_stride : +K1]] -> [groups, N, K0+K1]:
    the weight-gradient product in split precision with the operands converted on load (tsg_wgrad_f32s, include/tsg_hip.h).
    A [M, >= N], B0 [M, K0], B1 [M, >= K1] are fp32 matrices with contiguous rows (column slices are fine)."""
    require_device(A, B0, B1)
    A, B0 = _rows2d(A), _rows2d(B0)
    B1 = _rows2d(B1) if B1 is not None else None
    M, K0 = B0.shape
    N = A.shape[1] if N is None else N
    if A.shape[0] != M or (B1 is not None and B1.shape[0] != M) or (B1 is None) != (K1 == 0):
        raise ValueError("wgrad_f32s: operand rows differ")
    K = K0 + K1
    if out is not None:                        # written in place: [N, K] (one group) or [groups, N, K], rows ldc floats apart (a column slice is fine)
        if (out.dtype != torch.float32 or out.shape[-2:] != (N, K) or out.stride(-1) != 1 or out.stride(-2) % 4 or out.data_ptr() % 16
                or (out.dim() == 3) != (groups > 1) or (out.dim() == 3 and (out.shape[0] != groups or out.stride(0) % 4))):
            raise ValueError("wgrad_f32s: out must be an fp32 [N,K] / [groups,N,K] view with contiguous, 16-byte aligned rows")
        C, ldc, cgs = out, out.stride(-2), (out.stride(0) if out.dim() == 3 else 0)
    else:
        C = torch.empty(groups, N, K, device=A.device, dtype=torch.float32)
        ldc, cgs = K, N * K
    nb = int(load().tsg_wgrad_f32s_ws_bytes(M, N, K0, K1, groups))
    if nb < 0:
        raise ValueError(f"wgrad_f32s: unsupported shape M={M} N={N} K0={K0} K1={K1} groups={groups}")
    ws = torch.empty(nb, device=A.device, dtype=torch.uint8) if nb else None
    _call("tsg_wgrad_f32s", A, ptr(A), A.stride(0), a_group_stride, ptr(B0), B0.stride(0), K0,
          ptr(B1) if B1 is not None else None, B1.stride(0) if B1 is not None else 0, b1_group_stride, K1, shift, period,
          ptr(C), ldc, cgs, ptr(ws) if ws is not None else None, nb, M, N, groups)
    return C


def wgrad_f32s_out2(A: torch.Tensor, B0: torch.Tensor, B1: torch.Tensor, N: int, K1: int, a_group_stride: int, b1_group_stride: int,
                    shift: int, period: int):
    """``wgrad_f32s`` with two groups and TWO outputs (tsg_wgrad_f32s_out2): -> (C0 [2, N, K0], C1 [2, N, K1]) -- the LSTM layer's
    dW_ih and dW_hh of both directions from one launch, each in its parameter's shape (no slicing copies)."""
    require_device(A, B0, B1)
    A, B0, B1 = _rows2d(A), _rows2d(B0), _rows2d(B1)
    M, K0 = B0.shape
    C0 = torch.empty(2, N, K0, device=A.device, dtype=torch.float32)
    C1 = torch.empty(2, N, K1, device=A.device, dtype=torch.float32)
    nb = int(load().tsg_wgrad_f32s_ws_bytes(M, N, K0, K1, 2))
    if nb < 0:
        raise ValueError(f"wgrad_f32s_out2: unsupported shape M={M} N={N} K0={K0} K1={K1}")
    ws = torch.empty(nb, device=A.device, dtype=torch.uint8) if nb else None
    _call("tsg_wgrad_f32s_out2", A, ptr(A), A.stride(0), a_group_stride, ptr(B0), B0.stride(0), K0, ptr(B1), B1.stride(0), b1_group_stride, K1,
          shift, period, ptr(C0), K0, N * K0, ptr(C1), K1, N * K1, ptr(ws) if ws is not None else None, nb, M, N, 2)
    return C0, C1


def wgrad_bf16_out2(A: torch.Tensor, B0: torch.Tensor, B1: torch.Tensor, N: int, K1: int, a_group_stride: int, b1_group_stride: int,
                    shift: int, period: int):
    """``wgrad_f32s_out2`` for bf16 operands (tsg_wgrad_bf16_out2): -> (C0 [2, N, K0], C1 [2, N, K1]) fp32."""
    require_device(A, B0, B1)
    A, B0, B1 = _rows2d_bf(A), _rows2d_bf(B0), _rows2d_bf(B1)
    M, K0 = B0.shape
    C0 = torch.empty(2, N, K0, device=A.device, dtype=torch.float32)
    C1 = torch.empty(2, N, K1, device=A.device, dtype=torch.float32)
    nb = int(load().tsg_wgrad_f32s_ws_bytes(M, N, K0, K1, 2))
    if nb < 0:
        raise ValueError(f"wgrad_bf16_out2: unsupported shape M={M} N={N} K0={K0} K1={K1}")
    ws = torch.empty(nb, device=A.device, dtype=torch.uint8) if nb else None
    _call("tsg_wgrad_bf16_out2", A, ptr(A), A.stride(0), a_group_stride, ptr(B0), B0.stride(0), K0, ptr(B1), B1.stride(0), b1_group_stride, K1,
          shift, period, ptr(C0), K0, N * K0, ptr(C1), K1, N * K1, ptr(ws) if ws is not None else None, nb, M, N, 2)
    return C0, C1


def _rows2d_bf(t: torch.Tensor) -> torch.Tensor:
    """bf16 2-D operand whose rows are contiguous (a column slice of a row-major matrix is taken as it is)."""
    if t.dtype != torch.bfloat16 or t.dim() != 2:
        raise TypeError(f"bf16 matrix expected, got {t.dtype} {tuple(t.shape)}")
    return t if t.stride(1) == 1 and t.stride(0) % 4 == 0 and t.data_ptr() % 16 == 0 else t.contiguous()


def wgrad_bf16(A: torch.Tensor, B0: torch.Tensor, N: int = None, groups: int = 1, a_group_stride: int = 0,
               B1: torch.Tensor = None, K1: int = 0, b1_group_stride: int = 0, shift: int = 0, period: int = 0) -> torch.Tensor:
    """``wgrad_f32s`` for bf16 operands (tsg_wgrad_bf16): C[g] = A[:, g*a_group_stride : +N]^T @ [B0 | B1 shifted] in fp32, one bf16
    MFMA per product.  The weight gradients of the bf16 storage mode (the library's bf16 GEMM runs this contraction-major shape
    with a small output on 64 tiles: 187 us at [1024 x 16384] x [16384 x 1024])."""
    require_device(A, B0, B1)
    A, B0 = _rows2d_bf(A), _rows2d_bf(B0)
    B1 = _rows2d_bf(B1) if B1 is not None else None
    M, K0 = B0.shape
    N = A.shape[1] if N is None else N
    if A.shape[0] != M or (B1 is not None and B1.shape[0] != M) or (B1 is None) != (K1 == 0):
        raise ValueError("wgrad_bf16: operand rows differ")
    K = K0 + K1
    C = torch.empty(groups, N, K, device=A.device, dtype=torch.float32)
    nb = int(load().tsg_wgrad_f32s_ws_bytes(M, N, K0, K1, groups))
    if nb < 0:
        raise ValueError(f"wgrad_bf16: unsupported shape M={M} N={N} K0={K0} K1={K1} groups={groups}")
    ws = torch.empty(nb, device=A.device, dtype=torch.uint8) if nb else None
    _call("tsg_wgrad_bf16", A, ptr(A), A.stride(0), a_group_stride, ptr(B0), B0.stride(0), K0,
          ptr(B1) if B1 is not None else None, B1.stride(0) if B1 is not None else 0, b1_group_stride, K1, shift, period,
          ptr(C), K, N * K, ptr(ws) if ws is not None else None, nb, M, N, groups)
    return C


def _split_operand(m: torch.Tensor, k_dim: int, right: bool) -> torch.Tensor:
    """Split a (possibly transposed-view) fp32 matrix without materialising the transpose."""
    if m.is_contiguous():
        return split_bf16x3(m, k_dim, right)
    if m.t().is_contiguous():
        return split_bf16x3(m.t(), 1 - k_dim, right).t()
    return split_bf16x3(m.contiguous(), k_dim, right)


_OWN_GEMM = os.environ.get("TSG_GEMM", "1") != "0"          # A/B switch: 0 = operand planes + the library's bf16 GEMM everywhere
# The LSTM layers' three products (input projection, dX, [dW_ih | dW_hh]) on the hand-written split-on-load kernels (tsg_gemm_f32s,
# tsg_wgrad_f32s) instead of operand planes + the library's bf16 GEMM: the default since round 4.  In the step it is a tie in time
# (14.295 / 14.297 / 14.304 ms own vs 14.278 / 14.314 / 14.320 ms library, alternating processes on one box:
# profiles/r4/bench_lstm_own_gemm_ab_v1.txt) -- and it takes the operand-plane passes (1.59 -> 0.24 ms per step) and the library's bf16
# GEMMs (4.92 -> 0.60 ms) out of the step: all matrix work of the path is hand-written now.  TSG_LSTM_GEMM=lib keeps the old path (A/B).
_LSTM_OWN_GEMM = os.environ.get("TSG_LSTM_GEMM", "own") != "lib"
_LSTM_SPLITK = os.environ.get("TSG_LSTM_SPLITK", "1") != "0"          # A/B switch: the sentence LSTM's dX as 8 K-chunks (bmm + sum)
_OWN_TRANSPOSE = os.environ.get("TSG_TRANSPOSE", "own") != "torch"     # A/B switch: "torch" = .t().contiguous()
# Round 4 measured two copy eliminations in the step (profiles/r4/bench_no_copies_ab_v1.txt): the LSTM layer's dW_ih / dW_hh as two
# parameter-shaped outputs of ONE weight-gradient launch (tsg_wgrad_f32s_out2) -- adopted, it is what the layer runs; and the
# contraction-major GEMM operand (tsg_gemm_f32s_nn: dX = dY W with the weight as stored) instead of a transposed copy + tsg_gemm_f32s --
# SLOWER in the step (the transposing W stage costs each of the 13 GEMM launches more than the copy it deletes): the entry point and its
# wrapper (gemm_f32s_nn) remain, nothing here routes to it (round 5: the environment switch and its five dispatch sites are gone).


def gemm_f32s_ok(M: int, N: int, K: int) -> bool:
    """Shapes tsg_gemm_f32s takes AND is the faster path for (include/tsg_hip.h): whole tiles of 64 / 128 / 256 rows x 256 columns (the
    kernel picks the M tile).  Against the operand passes + the library's bf16 GEMM (tools/gemm_small_time.py, gemm_f32s_time.py):
    [2560 x 1024] x [1024 x 1024]^T 33 vs 42 us, [8192 x 1024] x [1024 x 1024]^T 57 vs 69, [16384 x 1024] x [4096 x 1024]^T 450 vs 427-479;
    with FEW tiles and a LONG contraction the library's split-K wins ([1280 x 4096] x [1024 x 4096]^T: 110 vs 80 us) and keeps the product."""
    if not (_OWN_GEMM and M % 64 == 0 and N % 256 == 0 and K % 32 == 0):
        return False
    return K <= 2048 or ((M + 255) // 256) * (N // 256) >= 100


def gemm_f32s(x: torch.Tensor, w: torch.Tensor, bias: torch.Tensor = None) -> torch.Tensor:
    """y [M,N] = x [M,K] @ w [N,K]^T (+ bias) in the split-precision arithmetic with the operands converted on load
    (tsg_gemm_f32s, csrc/gemm_f32s.hip): no operand planes in memory.  x, w fp32 contiguous."""
    require_device(x, w, bias)
    x, w = _f32c(x), _f32c(w)
    M, K = x.shape
    N = w.shape[0]
    if w.shape[1] != K:
        raise ValueError(f"gemm_f32s: x{tuple(x.shape)} w{tuple(w.shape)}")
    y = torch.empty(M, N, device=x.device, dtype=torch.float32)
    _call("tsg_gemm_f32s", x, ptr(x), ptr(w), ptr(_f32c(bias)) if bias is not None else None, ptr(y), M, N, K)
    return y


_AUTO = object()          # "the mode that is current now" (forward); a backward passes the mode its forward saved in ctx


def gemm_f32s_nn(x: torch.Tensor, w: torch.Tensor, w1: torch.Tensor = None) -> torch.Tensor:
    """y [M,N] = x [M,K] @ w [K,N] in the split-precision arithmetic with the RIGHT operand contraction-major (tsg_gemm_f32s_nn): the
    input gradient dX = dY W of a Linear with the weight as the parameter stores it -- no transposed copy.  w may be a column slice of a
    wider row-major matrix (row stride = its stride(0)); with ``w1`` the contraction rows are [w ; w1] (same row stride)."""
    require_device(x, w, w1)
    x = _f32c(x)
    M, K = x.shape
    k0, N = w.shape
    k1 = w1.shape[0] if w1 is not None else 0
    for t in (w, w1):
        if t is not None and (t.dtype != torch.float32 or t.stride(1) != 1 or t.stride(0) % 4 or t.data_ptr() % 16 or t.shape[1] != N
                              or t.stride(0) != w.stride(0)):
            raise ValueError("gemm_f32s_nn: the right operand must be fp32 row-major (or a column slice of such a matrix), 16-byte aligned rows")
    if k0 + k1 != K:
        raise ValueError(f"gemm_f32s_nn: x{tuple(x.shape)} w{tuple(w.shape)}")
    y = torch.empty(M, N, device=x.device, dtype=torch.float32)
    _call("tsg_gemm_f32s_nn", x, ptr(x), K, ptr(w), ptr(w1) if w1 is not None else None, k0, w.stride(0), None, ptr(y), N, M, N, K)
    return y


def gemm_f32s_nn_acc(x: torch.Tensor, w: torch.Tensor, y: torch.Tensor, w1: torch.Tensor = None) -> torch.Tensor:
    """y [M,N] += x [M,K] @ [w ; w1] [K,N] in the split-precision arithmetic (tsg_gemm_f32s_nn_acc): the input gradient dX = dY W of a Linear added
    onto a gradient buffer that another consumer of the same activation has already written -- no add kernel.  w (and w1: the contraction
    rows behind w's, same row stride) may be column slices of wider row-major parameters.  Returns y."""
    require_device(x, w, y, w1)
    x = _f32c(x)
    M, K = x.shape
    k0, N = w.shape
    k1 = w1.shape[0] if w1 is not None else 0
    for t in (w, w1):
        if t is not None and (t.dtype != torch.float32 or t.stride(1) != 1 or t.stride(0) % 4 or t.data_ptr() % 16 or t.shape[1] != N or t.stride(0) != w.stride(0)):
            raise ValueError("gemm_f32s_nn_acc: the right operand must be fp32 row-major (or a column slice of such a matrix), 16-byte aligned rows")
    if k0 + k1 != K or y.dtype != torch.float32 or y.shape != (M, N) or not y.is_contiguous():
        raise ValueError(f"gemm_f32s_nn_acc: x{tuple(x.shape)} w{tuple(w.shape)} y{tuple(y.shape)}")
    _call("tsg_gemm_f32s_nn_acc", x, ptr(x), K, ptr(w), ptr(w1) if w1 is not None else None, k0, w.stride(0), ptr(y), N, M, N, K)
    return y


def gemm_f32s_nn_ok(M: int, N: int, K: int) -> bool:
    """Shapes tsg_gemm_f32s_nn / _nn_acc take (include/tsg_hip.h)."""
    return _OWN_GEMM and M % 256 == 0 and N % 256 == 0 and K % 32 == 0


def _mm(a: torch.Tensor, b: torch.Tensor, mode=_AUTO) -> torch.Tensor:
    """fp32 [M,K] @ [K,N] -> fp32 in the GEMM precision ``mode`` (default: the configured one): rocBLAS / hipBLASLt, or -- "f32s", whole 256-tiles -- the
    hand-written split-on-load GEMM (the right operand is taken as [N,K] contiguous: a weight's `.t()` view as it is, a [K,N]
    contiguous matrix through one transposed copy, which for the path's weights is a few MB)."""
    mode = _GEMM_DTYPE if mode is _AUTO else mode
    if mode is None:
        return a @ b
    if mode == "f32s":
        if a.shape[1] % 4 or a.shape[0] % 4 or b.shape[1] % 4:
            return a @ b
        if a.is_cuda and a.is_contiguous() and gemm_f32s_ok(a.shape[0], b.shape[1], a.shape[1]):
            if b.t().is_contiguous():
                return gemm_f32s(a, b.t())
            if b.is_contiguous() and b.numel() <= (1 << 24):
                return gemm_f32s(a, transposed(b))
        return torch.mm(_split_operand(a, 1, False), _split_operand(b, 0, True), out_dtype=torch.float32)
    # bf16 operands, fp32 accumulate AND fp32 output straight from the GEMM (no bf16 round trip of the result, no cast kernel)
    return torch.mm(a.to(_BF), b.to(_BF), out_dtype=torch.float32)


_fwd = torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)     # under autocast: fp32 in, autocast off inside
_bwd = torch.amp.custom_bwd(device_type="cuda")


def _f32c(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32:
        raise TypeError(f"fp32 tensor expected, got {t.dtype}")
    return t.contiguous()


_BF = torch.bfloat16


def _bfc(t: torch.Tensor) -> torch.Tensor:
    """bf16 contiguous (activations of the bf16 storage mode; an fp32 tensor -- a model input -- is rounded once here)."""
    return t.to(_BF).contiguous()


_SHADOWS = os.environ.get("TSG_BF16_SHADOW", "1") != "0"       # 0: cast the fp32 parameters to bf16 at every use (A/B)


def weight_bf16(w: torch.Tensor) -> torch.Tensor:
    """The bf16 operand of an fp32 parameter in the bf16 storage mode.  For a leaf parameter the rounded copy is kept as a SHADOW on the parameter
    object and rewritten by the optimizer's own kernel with every update (engine.TsgAdam -> tsg_adam_step_shadow), so a training step does not
    cast its weights at all (eleven cast launches, 0.11 ms of the 7 ms GMD step).  The shadow is trusted while the parameter's version counter is
    the one it was made at: TsgAdam rewrites the shadow in its own kernel and re-stamps it (``restamp_shadow``); every other in-place change made
    THROUGH THE PARAMETER (``load_state_dict``, ``p.copy_()`` / ``p.mul_()`` under ``no_grad``, an optimizer that is not TsgAdam) bumps the version
    and the shadow is re-made here.  NOT seen: writes through ``p.data`` (``p.data.copy_()``, ``dist.broadcast(p.data)``, EMA weight swaps) or
    through raw pointers -- ``.data`` carries its own version counter -- so code that edits parameters that way calls
    ``invalidate_shadows(module)`` afterwards (dp.FlatGradAllReduce does, after its initial broadcast)."""
    wd = w.detach()
    if not (_SHADOWS and w.is_leaf and w.requires_grad and w.dtype == torch.float32 and w.is_cuda and w.is_contiguous()):
        return wd.to(_BF)
    sh = getattr(w, "_tsg_shadow", None)
    key = (w._version, w.data_ptr())                       # in-place edits bump the version; a re-pointed ``.data`` (``.to()``, a joined buffer) moves the address
    if sh is None or sh.shape != w.shape or sh.device != w.device:
        sh = wd.to(_BF)
        w._tsg_shadow, w._tsg_shadow_version = sh, key
    elif w._tsg_shadow_version != key:
        sh.copy_(wd)
        w._tsg_shadow_version = key
    return sh


def invalidate_shadows(module_or_params) -> int:
    """Drop the bf16 shadows of a module's (or an iterable's) parameters: the next bf16-storage forward re-makes them from the fp32 masters.
    Call it after editing parameters through ``p.data`` or raw pointers (see ``weight_bf16``); returns the number of shadows dropped."""
    ps = module_or_params.parameters() if hasattr(module_or_params, "parameters") else module_or_params
    n = 0
    for p in ps:
        if getattr(p, "_tsg_shadow", None) is not None:
            p._tsg_shadow = None
            p._tsg_shadow_version = None
            n += 1
    return n


def restamp_shadow(p: torch.Tensor) -> None:
    """The optimizer has rewritten ``p`` AND its shadow in one kernel through raw pointers: bump both version counters -- a backward that runs
    after the step on a graph that saved the old weights then fails autograd's saved-tensor check, as it does with torch's own in-place
    optimizers, instead of silently using the new weights -- and mark the shadow as current for the new version."""
    torch.autograd.graph.increment_version(p)
    sh = getattr(p, "_tsg_shadow", None)
    if sh is not None:
        torch.autograd.graph.increment_version(sh)
        p._tsg_shadow_version = (p._version, p.data_ptr())


def shadow_of(p: torch.Tensor):
    """The parameter's live bf16 shadow (``weight_bf16``) or None -- what the optimizer passes to tsg_adam_step_shadow."""
    sh = getattr(p, "_tsg_shadow", None)
    if sh is None or not _SHADOWS or getattr(p, "_tsg_shadow_version", None) != (p._version, p.data_ptr()) or sh.shape != p.shape or sh.device != p.device:
        return None
    return sh


def _f32p(t: torch.Tensor) -> torch.Tensor:
    """fp32 contiguous: parameters and the small [B,T] / [B,J] side inputs the kernels keep in fp32 in every mode."""
    return t.float().contiguous()


def _act(t: torch.Tensor, bf: bool) -> torch.Tensor:
    return _bfc(t) if bf else _f32c(t)


def scdm_bwd_fused_ok(B: int, T: int, N: int, H: int, Ds: int) -> bool:
    """True when the K1 / K1g backward of this shape runs as the one-launch kernel on the current device (the only backward the
    bf16 storage dtype takes): the library's own plan (tsg_scdm_bwd_fused_ok, include/tsg_hip.h), not a probe with a failing call."""
    return bool(load().tsg_scdm_bwd_fused_ok(B, T, N, H, Ds))


class _ScdmAttn(torch.autograd.Function):
    """K1: (a=[B,T,H], s=[B,N,H], w=[H], sent=[B,N,Ds]) -> (C=[B,T,Ds], P=[B,T,N])."""

    @staticmethod
    @_fwd
    def forward(ctx, a, s, w, sent):
        require_device(a, s, w, sent)
        bf = a.dtype == _BF                                    # bf16 storage: a, s, sent, C (and their gradients) as bf16
        a, s, w, sent = _act(a, bf), _act(s, bf), _f32p(w), _act(sent, bf)
        B, T, H = a.shape
        _, N, Ds = sent.shape
        if s.shape != (B, N, H) or w.numel() != H:
            raise ValueError(f"scdm_attn: shape mismatch a{tuple(a.shape)} s{tuple(s.shape)} w{tuple(w.shape)} sent{tuple(sent.shape)}")
        C = torch.empty(B, T, Ds, device=a.device, dtype=a.dtype)
        P = torch.empty(B, T, N, device=a.device, dtype=torch.float32)
        # "f32s" mode: TSG_F32S selects the forward whose P @ sent product runs as split-precision bf16 MFMAs (csrc/scdm_attn.hip)
        ctx.dt = TSG_BF16 if bf else (TSG_F32S if _GEMM_DTYPE == "f32s" else TSG_F32)
        _call("tsg_scdm_attn_fwd", a, ptr(a), ptr(s), ptr(w), ptr(sent), ptr(C), ptr(P),
                                       B, T, N, H, Ds, ctx.dt)
        ctx.save_for_backward(a, s, w, sent, P)
        ctx.mark_non_differentiable(P)
        return C, P

    @staticmethod
    @_bwd
    def backward(ctx, dC, _dP):
        a, s, w, sent, P = ctx.saved_tensors
        dC = _act(dC, ctx.dt == TSG_BF16)
        B, T, H = a.shape
        _, N, Ds = sent.shape
        da = torch.empty_like(a); ds = torch.empty_like(s)
        dw = torch.empty_like(w); dsent = torch.empty_like(sent)
        nb = int(load().tsg_scdm_bwd_ws_bytes(B, T, N, H, Ds, 0))
        ws = torch.empty(nb // 4 + 4, device=a.device, dtype=torch.float32)
        if ctx.dt != TSG_BF16 or scdm_bwd_fused_ok(B, T, N, H, Ds):
            _gate_before()                          # the partner exchange of the one-launch backward wants its workgroup pairs co-resident
            _call("tsg_scdm_attn_bwd", a, ptr(a), ptr(s), ptr(w), ptr(sent), ptr(P), ptr(dC), ptr(da), ptr(ds),
                                           ptr(dw), ptr(dsent), ptr(ws), nb, B, T, N, H, Ds, ctx.dt)
        else:
            # a shape only the two-kernel (fp32-storage) backward takes (decided by the library's predicate, not by a failing
            # call): fp32 copies through it, results rounded to bf16
            af, sf, vf, gf = a.float(), s.float(), sent.float(), dC.float()
            daf, dsf, dvf = torch.empty_like(af), torch.empty_like(sf), torch.empty_like(vf)
            _call("tsg_scdm_attn_bwd", af, ptr(af), ptr(sf), ptr(w), ptr(vf), ptr(P), ptr(gf), ptr(daf), ptr(dsf),
                                           ptr(dw), ptr(dvf), ptr(ws), nb, B, T, N, H, Ds, TSG_F32)
            da, ds, dsent = daf.to(_BF), dsf.to(_BF), dvf.to(_BF)
        return da, ds, dw, dsent


def scdm_attn(a, s, w, sent, return_p: bool = False):
    """Fused SCDM additive cross-attention on projected inputs (see include/tsg_hip.h, K1)."""
    C, P = _ScdmAttn.apply(a, s, w.reshape(-1), sent)      # [1,H] Linear weight or [H]: autograd undoes the view
    return (C, P) if return_p else C


class _ScdmGate(torch.autograd.Function):
    """K1g: (a, s, w, VW=[B,N,Ds], gbias=[Ds], r=[B,T,Ds]) -> out = r * sigmoid(P @ VW + gbias)."""

    @staticmethod
    @_fwd
    def forward(ctx, a, s, w, VW, gbias, r, sink=None):
        require_device(a, s, w, VW, gbias, r)
        ctx.sink, ctx.r_in = sink, (r if sink is not None else None)
        bf = r.dtype == _BF                                    # bf16 storage: a, s, VW, r, out (and their gradients) as bf16
        a, s, w, VW, gbias, r = _act(a, bf), _act(s, bf), _f32p(w), _act(VW, bf), _f32p(gbias), _act(r, bf)
        B, T, H = a.shape
        _, N, Ds = VW.shape
        if s.shape != (B, N, H) or w.numel() != H or r.shape != (B, T, Ds) or gbias.numel() != Ds:
            raise ValueError(f"scdm_gate: shape mismatch a{tuple(a.shape)} s{tuple(s.shape)} VW{tuple(VW.shape)} r{tuple(r.shape)}")
        out = torch.empty(B, T, Ds, device=a.device, dtype=a.dtype)
        P = torch.empty(B, T, N, device=a.device, dtype=torch.float32)
        ctx.dt = TSG_BF16 if bf else (TSG_F32S if _GEMM_DTYPE == "f32s" else TSG_F32)
        _call("tsg_scdm_gate_fwd", a, ptr(a), ptr(s), ptr(w), ptr(VW), ptr(gbias), ptr(r), ptr(out), ptr(P),
              B, T, N, H, Ds, ctx.dt)
        ctx.save_for_backward(a, s, w, VW, gbias, r, P)
        return out

    @staticmethod
    @_bwd
    def backward(ctx, dout):
        a, s, w, VW, gbias, r, P = ctx.saved_tensors
        dout = _act(dout, ctx.dt == TSG_BF16)
        B, T, H = a.shape
        _, N, Ds = VW.shape
        da = torch.empty_like(a); ds = torch.empty_like(s); dw = torch.empty_like(w)
        dVW = torch.empty_like(VW); dgb = torch.empty_like(gbias); dr = torch.empty_like(r)
        nb = int(load().tsg_scdm_bwd_ws_bytes(B, T, N, H, Ds, 1))
        ws = torch.empty(nb // 4 + 4, device=a.device, dtype=torch.float32)
        if ctx.dt != TSG_BF16 or scdm_bwd_fused_ok(B, T, N, H, Ds):
            _gate_before()                          # the partner exchange of the one-launch backward wants its workgroup pairs co-resident
            _call("tsg_scdm_gate_bwd", a, ptr(a), ptr(s), ptr(w), ptr(VW), ptr(gbias), ptr(r), ptr(P), ptr(dout),
                  ptr(da), ptr(ds), ptr(dw), ptr(dVW), ptr(dgb), ptr(dr), ptr(ws), nb, B, T, N, H, Ds, ctx.dt)
        else:
            af, sf, vf, rf, gf = a.float(), s.float(), VW.float(), r.float(), dout.float()
            daf, dsf, dvf, drf = torch.empty_like(af), torch.empty_like(sf), torch.empty_like(vf), torch.empty_like(rf)
            _call("tsg_scdm_gate_bwd", af, ptr(af), ptr(sf), ptr(w), ptr(vf), ptr(gbias), ptr(rf), ptr(P), ptr(gf),
                  ptr(daf), ptr(dsf), ptr(dw), ptr(dvf), ptr(dgb), ptr(drf), ptr(ws), nb, B, T, N, H, Ds, TSG_F32)
            da, ds, dVW, dr = daf.to(_BF), dsf.to(_BF), dvf.to(_BF), drf.to(_BF)
        if ctx.sink is not None and ctx.sink.tensor is not None and dr.dtype == ctx.sink.tensor.dtype:
            _sink_add(ctx.sink, ctx.r_in, dr)                  # r is a shared activation (the BiLSTM output: also W_a's input): its gradient meets the other there
            dr = None
        return da, ds, dw, dVW, dgb, dr, None


def scdm_gate(a, s, w, VW, gbias, r):
    """SCDM attention fused with the recalibration gate (see include/tsg_hip.h, K1g)."""
    return _ScdmGate.apply(a, s, w.reshape(-1), VW, gbias, r, _sink_for(r))


class _ScdmGateProj(torch.autograd.Function):
    """The recalibration block's tail on ONE autograd node (f32s mode): a = x W_a^T, out = x * sigmoid(P(a, s) @ VW + gbias) -- the BiLSTM output x
    is both the attention's clip operand (through W_a) and the gate's r (reference VideoEncoder.py:52-59, attention.py:104-121).  As two nodes
    (``linear`` + ``scdm_gate``) autograd adds their two gradients of x with an elementwise kernel over [B, T, D] (30 us at [128, 128, 1024], twice
    per GMD step); here the K1g backward writes dr and the input-gradient GEMM of W_a adds da W_a onto that buffer in its epilogue
    (tsg_gemm_f32s_nn_acc).  Same kernels, same arithmetic: the sum is the same single fp32 addition per element."""

    @staticmethod
    @_fwd
    def forward(ctx, x, wa, s, w, VW, gbias):
        require_device(x, wa, s, w, VW, gbias)
        x, wa, s, w, VW, gbias = _f32c(x), _f32c(wa), _f32c(s), _f32p(w), _f32c(VW), _f32p(gbias)
        B, T, Dv = x.shape
        H = wa.shape[0]
        _, N, Ds = VW.shape
        if wa.shape[1] != Dv or s.shape != (B, N, H) or w.numel() != H or Ds != Dv or gbias.numel() != Ds:
            raise ValueError(f"scdm_gate_proj: shape mismatch x{tuple(x.shape)} W_a{tuple(wa.shape)} s{tuple(s.shape)} VW{tuple(VW.shape)}")
        a = gemm_f32s(x.view(B * T, Dv), wa).view(B, T, H)
        out = torch.empty(B, T, Ds, device=x.device, dtype=torch.float32)
        P = torch.empty(B, T, N, device=x.device, dtype=torch.float32)
        _call("tsg_scdm_gate_fwd", a, ptr(a), ptr(s), ptr(w), ptr(VW), ptr(gbias), ptr(x), ptr(out), ptr(P), B, T, N, H, Ds, TSG_F32S)
        ctx.save_for_backward(x, wa, a, s, w, VW, gbias, P)
        return out

    @staticmethod
    @_bwd
    def backward(ctx, dout):
        x, wa, a, s, w, VW, gbias, P = ctx.saved_tensors
        dout = _f32c(dout)
        B, T, Dv = x.shape
        H = wa.shape[0]
        _, N, Ds = VW.shape
        da = torch.empty_like(a); ds = torch.empty_like(s); dw = torch.empty_like(w)
        dVW = torch.empty_like(VW); dgb = torch.empty_like(gbias); dx = torch.empty_like(x)
        nb = int(load().tsg_scdm_bwd_ws_bytes(B, T, N, H, Ds, 1))
        ws = torch.empty(nb // 4 + 4, device=x.device, dtype=torch.float32)
        _gate_before()
        _call("tsg_scdm_gate_bwd", a, ptr(a), ptr(s), ptr(w), ptr(VW), ptr(gbias), ptr(x), ptr(P), ptr(dout),
              ptr(da), ptr(ds), ptr(dw), ptr(dVW), ptr(dgb), ptr(dx), ptr(ws), nb, B, T, N, H, Ds, TSG_F32S)      # dx <- dr
        da2, x2 = da.view(B * T, H), x.view(B * T, Dv)
        dwa = None
        if ctx.needs_input_grad[1]:
            dwa = wgrad_f32s(da2, x2)[0] if _WGRAD_KERNEL and wgrad_f32s_ok(B * T, H, Dv) else da2.t() @ x2
        if ctx.needs_input_grad[0]:
            gemm_f32s_nn_acc(da2, wa, dx.view(B * T, Dv))                                                           # dx += da W_a
        else:
            dx = None
        return dx, dwa, ds, dw.view_as(w), dVW, dgb


def scdm_gate_proj_ok(x: torch.Tensor, wa: torch.Tensor, VW: torch.Tensor) -> bool:
    """The one-node form applies: f32s mode on the GPU, fp32 tensors, whole GEMM tiles, the gate as wide as the clip features."""
    if not (_GEMM_DTYPE == "f32s" and x.is_cuda and x.dtype == torch.float32 and not bf16_storage() and x.dim() == 3 and not torch.is_autocast_enabled()):
        return False
    if not _SHARED_GRAD:
        return False
    M, Dv, H = x.shape[0] * x.shape[1], x.shape[2], wa.shape[0]
    return VW.shape[-1] == Dv and gemm_f32s_ok(M, H, Dv) and gemm_f32s_nn_ok(M, Dv, H) and M >= 2048


def scdm_gate_proj(x, wa, s, w, VW, gbias):
    """out = x * sigmoid(softmax_n(w . tanh(W_s s_n + W_a x_t + b)) @ VW + gbias) with the clip projection inside the node (``_ScdmGateProj``)."""
    return _ScdmGateProj.apply(x, wa, s, w.reshape(-1), VW, gbias)


_k3_ws = {}        # (device, stream, B, T, Hm) -> uint8 workspace of the one-launch K3 backward; its ticket counters are zero between calls


def _k3_workspace(device, B: int, T: int, Hm: int):
    """Workspace of tsg_boundary_score_bwd_ws (include/tsg_hip.h): [roundup(B,4) ticket counters | partial rows], zeroed when created,
    its counters left zeroed by every call.  That invariant holds per LAYOUT only -- with another B the counter words of one call
    overlap the partial rows of another (B=64,T=128 then B=128,T=32 left counters 64..127 non-zero and dgate unwritten)
    -- so the cache is keyed on the shape as well as on (device, stream).  A buffer created while the stream is being captured belongs
    to the graph's pool and is zeroed by a captured fill on every replay: it is handed out once and not kept."""
    nb = int(load().tsg_boundary_score_bwd_ws_bytes(B, T, Hm))
    if nb <= 0:
        raise ValueError(f"boundary_score: bad shape B={B} T={T} Hm={Hm}")
    key = (_cuda_device(device), torch.cuda.current_stream(device).cuda_stream, B, T, Hm)
    ws = _k3_ws.get(key)
    if ws is None or ws.numel() < nb:
        ws = torch.zeros(nb, device=device, dtype=torch.uint8)
        if not torch.cuda.is_current_stream_capturing():     # (the warm-up steps before a capture run on the capture stream: kept)
            if len(_k3_ws) >= 64:                            # bounded: a sweep over many shapes does not pile buffers up
                _k3_ws.clear()
            _k3_ws[key] = ws
    return ws, nb


def _k3_call(name: str, like: torch.Tensor, *args) -> None:
    """``_call`` for the entry points that take a cached K3 workspace: when the call does not return 0 the cache is dropped, so that no
    later call trusts tickets a failed call may have touched (round-4 review: "left zeroed by every call" only holds for calls that
    ran to their end)."""
    try:
        _call(name, like, *args)
    except Exception:
        _k3_ws.clear()
        raise


class _BoundaryScore(torch.autograd.Function):
    """K3: (y=[B,T,2Hm], cs=[B,2Hm], b1=[2Hm], w2=[2Hm], b2=[2], gate=[B,T]|None, mask=[B,T] int|None)
    -> (p_start, p_end) [B,T]."""

    @staticmethod
    @_fwd
    def forward(ctx, y, cs, b1, w2, b2, gate, mask):
        require_device(y, cs, b1, w2, b2, gate, mask)
        bf = y.dtype == _BF                                    # bf16 storage: y / dy as bf16, everything [B,T]- or [B,J]-sized fp32
        y, cs, b1, w2, b2 = _act(y, bf), _f32p(cs), _f32p(b1), _f32p(w2), _f32p(b2)
        ctx.dt = TSG_BF16 if bf else TSG_F32
        B, T, J = y.shape
        if J % 2 or cs.shape != (B, J) or b1.numel() != J or w2.numel() != J or b2.numel() != 2:
            raise ValueError(f"boundary_score: shape mismatch y{tuple(y.shape)} cs{tuple(cs.shape)}")
        gate_c = _f32p(gate) if gate is not None else None
        mask_c = mask.to(torch.int32).contiguous() if mask is not None else None
        ps = torch.empty(B, T, device=y.device, dtype=torch.float32)
        pe = torch.empty_like(ps)
        _call("tsg_boundary_score_fwd", y, ptr(y), ptr(cs), ptr(b1), ptr(w2), ptr(b2),
                                            ptr(gate_c) if gate_c is not None else None,
                                            ptr(mask_c) if mask_c is not None else None,
                                            ptr(ps), ptr(pe), B, T, J // 2, ctx.dt)
        ctx.save_for_backward(y, cs, b1, w2, ps, pe, *( [gate_c] if gate_c is not None else []), *([mask_c] if mask_c is not None else []))
        ctx.has_gate, ctx.has_mask = gate_c is not None, mask_c is not None
        return ps, pe

    @staticmethod
    @_bwd
    def backward(ctx, dps, dpe):
        saved = list(ctx.saved_tensors)
        y, cs, b1, w2, ps, pe = saved[:6]
        rest = saved[6:]
        gate = rest.pop(0) if ctx.has_gate else None
        mask = rest.pop(0) if ctx.has_mask else None
        B, T, J = y.shape
        dps = _f32p(dps) if dps is not None else torch.zeros_like(ps)
        dpe = _f32p(dpe) if dpe is not None else torch.zeros_like(pe)
        dy = torch.empty_like(y); dcs = torch.empty_like(cs)
        db1p = torch.empty(B, J, device=y.device, dtype=torch.float32)
        dw2p = torch.empty_like(db1p)
        db2p = torch.empty(B, 2, device=y.device, dtype=torch.float32)
        dgate = torch.empty(B, T, device=y.device, dtype=torch.float32) if gate is not None else None
        ws, nb = _k3_workspace(y.device, B, T, J // 2)
        _k3_call("tsg_boundary_score_bwd_ws", y, ptr(y), ptr(cs), ptr(b1), ptr(w2),
                                               ptr(gate) if gate is not None else None,
                                               ptr(mask) if mask is not None else None,
                                               ptr(ps), ptr(pe), ptr(dps), ptr(dpe), ptr(dy), ptr(dcs), ptr(db1p),
                                               ptr(dw2p), ptr(db2p), ptr(dgate) if dgate is not None else None, ptr(ws), nb,
                                               B, T, J // 2, ctx.dt)
        return dy, dcs, db1p.sum(0), dw2p.sum(0), db2p.sum(0), dgate, None


def boundary_score(y, cs, b1, w2, b2, gate=None, mask=None):
    """Fused boundary head after the video-half GEMM (see include/tsg_hip.h, K3)."""
    return _BoundaryScore.apply(y, cs, b1, w2, b2, gate, mask)


class _MHA(torch.autograd.Function):
    """K2: (Q=[B,Tq,dk], K=[B,Tk,dk], V=[B,Tk,dv]) -> (O=[B,Tq,dv], A_sum, S_sum) with the reference's
    chunked heads and caller-provided scale divisor."""

    @staticmethod
    @_fwd
    def forward(ctx, Q, K, V, n_heads, scale, causal, want_maps, p_drop=0.0, seed=0, offset=0, rng=None):
        require_device(Q, K, V)
        bf = Q.dtype == _BF                                    # bf16 storage: Q, K, V, O and their gradients as bf16 (TSG_BF16)
        Q, K, V = _act(Q, bf), _act(K, bf), _act(V, bf)
        B, Tq, dk = Q.shape
        _, Tk, dv = V.shape
        if K.shape != (B, Tk, dk):
            raise ValueError(f"mha: shape mismatch Q{tuple(Q.shape)} K{tuple(K.shape)} V{tuple(V.shape)}")
        if causal and Tq != Tk:
            raise ValueError("mha: causal attention needs Tq == Tk (the reference subtracts a [Tk,Tk] triangle)")
        O = torch.empty(B, Tq, dv, device=Q.device, dtype=Q.dtype)
        lse = torch.empty(B, n_heads, Tq, device=Q.device, dtype=torch.float32)
        A = torch.empty(B, Tq, Tk, device=Q.device, dtype=torch.float32) if want_maps else None
        S = torch.empty(B, Tq, Tk, device=Q.device, dtype=torch.float32) if want_maps else None
        head = (ptr(Q), ptr(K), ptr(V), ptr(O), ptr(A) if want_maps else None, ptr(S) if want_maps else None, ptr(lse), B, Tq, Tk, dk, dv,
                int(n_heads), float(scale), int(bool(causal)), float(p_drop))
        # in the split-precision GEMM mode the attention products run on the bf16 MFMA as hi/lo products as well (TSG_F32S:
        # include/tsg_hip.h, K2; shapes the split kernels do not cover, and the A_forward maps, run the exact kernels)
        dt = TSG_BF16 if bf else (TSG_F32S if _GEMM_DTYPE in ("f32s", "bf16") else TSG_F32)
        if rng is not None:                                         # (seed, offset) in device memory: graph-capture safe
            _call("tsg_mha_fwd_rng", Q, *head, ptr(rng), dt)
        else:
            _call("tsg_mha_fwd", Q, *head, int(seed), int(offset), dt)
        ctx.save_for_backward(Q, K, V, O, lse, *([rng] if rng is not None else []))
        ctx.cfg = (int(n_heads), float(scale), int(bool(causal)))
        ctx.drop = (float(p_drop), int(seed), int(offset))          # the backward regenerates the same mask
        ctx.has_rng = rng is not None
        ctx.bwd_dtype = dt
        ctx.set_materialize_grads(False)                            # no zero tensors for the non-differentiable maps' grads
        if want_maps:
            ctx.mark_non_differentiable(A, S)
            return O, A, S
        return O, None, None

    @staticmethod
    @_bwd
    def backward(ctx, dO, _dA, _dS):
        Q, K, V, O, lse = ctx.saved_tensors[:5]
        rng = ctx.saved_tensors[5] if ctx.has_rng else None
        n_heads, scale, causal = ctx.cfg
        p_drop, seed, offset = ctx.drop
        dO = _act(dO, ctx.bwd_dtype == TSG_BF16) if dO is not None else torch.zeros_like(O)
        B, Tq, dk = Q.shape
        _, Tk, dv = V.shape
        dQ = torch.empty_like(Q); dK = torch.empty_like(K); dV = torch.empty_like(V)
        delta = torch.empty_like(lse)
        head = (ptr(Q), ptr(K), ptr(V), ptr(O), ptr(dO), ptr(lse), ptr(dQ), ptr(dK), ptr(dV), ptr(delta), B, Tq, Tk, dk, dv, n_heads, scale,
                causal, p_drop)
        if rng is not None:
            _call("tsg_mha_bwd_rng", Q, *head, ptr(rng), ctx.bwd_dtype)
        else:
            _call("tsg_mha_bwd", Q, *head, seed, offset, ctx.bwd_dtype)
        return dQ, dK, dV, None, None, None, None, None, None, None, None


def mha_bf16_ok(d_key: int, d_value: int, n_heads: int, want_maps: bool = False) -> bool:
    """Shapes tsg_mha_fwd AND tsg_mha_bwd take with dtype TSG_BF16 (include/tsg_hip.h): equal key / value widths, head widths
    32 .. 128 in steps of 32, no A_forward maps."""
    dh = d_key // max(n_heads, 1)
    return (not want_maps) and d_key == d_value and d_key % max(n_heads, 1) == 0 and dh % 32 == 0 and 32 <= dh <= 128


_mha_rng_state = {}        # device -> int64 [2] tensor (seed, offset): the dropout counter of captured launches


def mha(Q, K, V, n_heads, scale, causal=False, return_maps=False, p_drop=0.0):
    """Fused multi-head attention on projected inputs (see include/tsg_hip.h, K2).
    Returns O, or (O, A_sum, S_sum) with ``return_maps`` (S_sum is the un-dropped softmax, as in the reference).
    ``p_drop`` > 0 applies attention dropout inside the kernel (out = dropout(softmax) V, attention.py:53-54): the mask
    is a counter-based hash of (seed, offset, element index); seed = torch.initial_seed(), offset drawn from torch's CPU
    generator, so ``torch.manual_seed`` makes it reproducible and every call gets a fresh mask.  While the stream is being
    captured into a HIP graph the (seed, offset) pair lives in device memory instead: the captured increment of the offset
    makes every REPLAY draw a fresh mask (a host-side offset would be frozen into the graph)."""
    seed = offset = 0
    rng = None
    if p_drop > 0.0:
        seed = torch.initial_seed() & 0xFFFFFFFFFFFFFFFF
        if Q.is_cuda and torch.cuda.is_current_stream_capturing():
            st = _mha_rng_state.get(_cuda_device(Q.device))
            if st is None:
                raise RuntimeError("mha: attention dropout under graph capture needs functional.mha_graph_rng(device) called before the "
                                   "capture (engine.GraphedTrainStep does)")
            st[1] += 1                                              # captured: advances on every replay
            rng = st.clone()                                        # this call's (seed, offset), shared by forward and backward
        else:
            offset = int(torch.randint(0, 2 ** 62, (1,)).item())
    if Q.dtype == _BF and not mha_bf16_ok(Q.shape[-1], V.shape[-1], n_heads, return_maps):
        # bf16 storage mode, a shape the TSG_BF16 kernels do not take (A_forward maps, head widths that are not multiples of 32 or
        # above 128, d_key != d_value): the fp32-storage kernels on fp32 copies, O rounded back
        O, A, S = _MHA.apply(Q.float(), K.float(), V.float(), n_heads, scale, causal, return_maps, float(p_drop), seed, offset, rng)
        return (O.to(_BF), A, S) if return_maps else O.to(_BF)
    O, A, S = _MHA.apply(Q, K, V, n_heads, scale, causal, return_maps, float(p_drop), seed, offset, rng)
    return (O, A, S) if return_maps else O


def _cuda_device(device) -> torch.device:
    """``device`` with its index resolved: torch.device('cuda') and torch.device('cuda', current) are different dict keys."""
    device = torch.device(device)
    if device.type == "cuda" and device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    return device


def mha_graph_rng(device):
    """Create (outside any capture) the device-resident (seed, offset) state the in-kernel attention dropout uses while a HIP
    graph is being captured / replayed on ``device``."""
    device = _cuda_device(device)
    st = _graph_keys.get(device)
    if st is None or st[0].shape[0] - st[1] < 256:           # (re)filled here, outside any capture: a capture could only allocate from its own pool
        _graph_keys[device] = [torch.zeros(kKeySlots, 2, dtype=torch.int32, device=device), 0]       # key slots of the captured dropout calls (see
        _graph_key_buffers.append(_graph_keys[device][0])                                             # _graph_key_slot); a replaced buffer stays alive for
                                                                                                      # the graphs captured on it (they replay its addresses)
    if device not in _mha_rng_state:
        seed = torch.initial_seed() & 0x7FFFFFFFFFFFFFFF
        _mha_rng_state[device] = torch.tensor([seed, int(torch.randint(0, 2 ** 40, (1,)).item())], dtype=torch.int64, device=device)
    return _mha_rng_state[device]


class _LinearHip(torch.autograd.Function):
    """y = x W^T (+ b) through the hand-written fp32 MFMA GEMM (tsg_linear_fwd); the two backward products reuse it
    (dx = dy W as a Linear with the transposed weight) or go to rocBLAS (dW = dy^T x: contraction over the rows)."""

    @staticmethod
    @_fwd
    def forward(ctx, x, w, b):
        require_device(x, w)
        x2, w = _f32c(x.reshape(-1, x.shape[-1])), _f32c(w)
        M, K = x2.shape
        N = w.shape[0]
        if w.shape[1] != K or K % 4:
            raise ValueError(f"linear_hip: x{tuple(x.shape)} w{tuple(w.shape)} (K must match and be a multiple of 4)")
        y = torch.empty(M, N, device=x.device, dtype=torch.float32)
        _call("tsg_linear_fwd", x2, ptr(x2), ptr(w), ptr(_f32c(b)) if b is not None else None, ptr(y), M, N, K, TSG_F32)
        ctx.save_for_backward(x2, w)
        ctx.has_bias, ctx.xshape = b is not None, x.shape
        return y.view(*x.shape[:-1], N)

    @staticmethod
    @_bwd
    def backward(ctx, dy):
        x2, w = ctx.saved_tensors
        dy2 = _f32c(dy.reshape(-1, dy.shape[-1]))
        M, K = x2.shape
        N = w.shape[0]
        dx = None
        if ctx.needs_input_grad[0]:
            if N % 4 == 0:
                wt = transposed(w)                                       # [K, N]: dx = dy W = Linear(dy, W^T)
                dx = torch.empty(M, K, device=dy.device, dtype=torch.float32)
                _call("tsg_linear_fwd", dy2, ptr(dy2), ptr(wt), None, ptr(dx), M, K, N, TSG_F32)
            else:
                dx = dy2 @ w
            dx = dx.view(ctx.xshape)
        dw = dy2.t() @ x2 if ctx.needs_input_grad[1] else None
        db = dy2.sum(0) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        return dx, dw, db


class _MatchHead(torch.autograd.Function):
    """logits[b,t] = w2 . act(y[b,t,:] + cs[b,:]) + b2  (K5, csrc/match_head.hip)."""

    @staticmethod
    @_fwd
    def forward(ctx, y, cs, w2, b2, act):
        require_device(y, cs, w2, b2)
        bf = y.dtype == _BF                                    # bf16 storage: y / dy as bf16
        y, cs, w2, b2 = _act(y, bf), _f32p(cs), _f32p(w2).view(-1), _f32p(b2).view(-1)
        ctx.dt = TSG_BF16 if bf else TSG_F32
        B, T, H = y.shape
        if cs.shape != (B, H) or w2.numel() != H or b2.numel() != 1:
            raise ValueError(f"match_head: shape mismatch y{tuple(y.shape)} cs{tuple(cs.shape)} w2{tuple(w2.shape)} b2{tuple(b2.shape)}")
        out = torch.empty(B, T, device=y.device, dtype=torch.float32)
        _call("tsg_match_head_fwd", y, ptr(y), ptr(cs), ptr(w2), ptr(b2), ptr(out), B, T, H, int(act), ctx.dt)
        ctx.save_for_backward(y, cs, w2)
        ctx.act = int(act)
        return out

    @staticmethod
    @_bwd
    def backward(ctx, dl):
        y, cs, w2 = ctx.saved_tensors
        B, T, H = y.shape
        dl = _f32p(dl)
        dy = torch.empty_like(y)
        dcs = torch.empty_like(cs); dw2 = torch.empty_like(w2); db2 = torch.empty(1, device=y.device, dtype=torch.float32)
        _call("tsg_match_head_bwd", y, ptr(y), ptr(cs), ptr(w2), ptr(dl), ptr(dy), ptr(dcs), ptr(dw2), ptr(db2), B, T, H, ctx.act, ctx.dt)
        return dy, dcs, dw2, db2, None


_ACTS = {"relu": 0, "tanh": 1, "sigmoid": 2}


def match_head(y, cs, w2, b2, activation="relu"):
    """Matching-head tail (include/tsg_hip.h, K5): y [B,T,H] video half of the first Linear, cs [B,H] query half + bias,
    w2 [H] / b2 [1] the 1-output second Linear -> raw matching logits [B,T]."""
    return _MatchHead.apply(y, cs, w2.reshape(-1), b2.reshape(-1), _ACTS[activation])


def head_gemm_ok(M: int, N: int, K: int, T: int, head_width: int) -> bool:
    """Shapes the fused head GEMMs (tsg_match_head_gemm / tsg_boundary_head_gemm, include/tsg_hip.h) take; the modules use them in the
    "f32s" mode (their arithmetic) and keep GEMM + K3 / K5 otherwise."""
    return (_GEMM_DTYPE == "f32s" and _OWN_GEMM and M > 0 and M % 64 == 0 and N % 256 == 0 and head_width % 256 == 0 and K % 32 == 0
            and 0 < T <= 8192 and M % T == 0 and M <= (1 << 22))


class GradSink:
    """The gradient of ONE activation with several consumers (the final clip features of GMD: matching head, boundary head on the leading rows,
    temporal-order discriminator -- SpanGroundMatchDisc.py:80-96), summed inside the consumers' own kernels: the first consumer to run its
    backward leaves its input gradient here, the later ones ADD theirs in the epilogue of their input-gradient GEMM (tsg_gemm_f32s_nn_acc), all
    of them return None for the activation, and ``_ShareGrad.backward`` hands the sum upstream.  As separate gradients autograd summed them with
    elementwise kernels (and a zero-filled full-size buffer + copy for the row slice): 85 us of a 12.2 ms step."""
    __slots__ = ("tensor", "buf")

    def __init__(self, tensor):
        self.tensor, self.buf = tensor, None

    def rows(self, x):
        """Row range of ``x`` (the shared activation or a contiguous block of its leading-axis rows) inside the shared tensor."""
        t = self.tensor
        row_bytes = t.stride(0) * t.element_size()
        r0 = (x.data_ptr() - t.data_ptr()) // row_bytes
        return r0, r0 + x.shape[0]

    def take(self, dx):
        """First contribution, full size: the sink adopts the tensor."""
        self.buf = dx

    def full(self):
        """The [rows, D] buffer for an accumulating contribution (zero-filled if nobody has written yet)."""
        if self.buf is None:
            self.buf = torch.zeros_like(self.tensor)
        return self.buf


_ACTIVE_SINK = None                # (module state, like the precision mode: one training thread per process -- one process per GPU)
_SHARED_GRAD = os.environ.get("TSG_SHARED_GRAD", "1") != "0"      # 0: autograd sums the gradients of shared activations itself (A/B)


class _ShareGrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, sink):
        ctx.sink = sink
        ctx.set_materialize_grads(False)
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        buf = ctx.sink.buf
        ctx.sink.buf = ctx.sink.tensor = None
        if buf is None:
            return g, None
        return (buf if g is None else buf + g), None


class shared_grad:
    """``with shared_grad(x) as xs:`` -- consumers called on ``xs`` (or on a block of its leading-axis rows) inside the block that know about sinks
    (match_head_params, boundary_head_params, moment_pool) sum their input gradients in place (``GradSink``).  Outside the f32s GPU training path
    the context is a no-op and yields ``x`` itself."""

    def __init__(self, x):
        mode_ok = (x.dtype == torch.float32 and _GEMM_DTYPE == "f32s" and not bf16_storage()) or (x.dtype == _BF and bf16_storage())
        ok = (_SHARED_GRAD and x.is_cuda and mode_ok and x.dim() == 3 and x.is_contiguous() and x.requires_grad
              and torch.is_grad_enabled() and not torch.is_autocast_enabled())
        self.x, self.sink = x, None
        if ok:
            self.sink = GradSink(None)
            self.x = _ShareGrad.apply(x, self.sink)
            self.sink.tensor = self.x

    def __enter__(self):
        global _ACTIVE_SINK
        self.prev, _ACTIVE_SINK = _ACTIVE_SINK, self.sink
        return self.x

    def __exit__(self, *exc):
        global _ACTIVE_SINK
        _ACTIVE_SINK = self.prev
        return False


def _sink_for(x):
    """The active sink if ``x`` is its tensor or a contiguous block of its rows (same trailing shape), else None."""
    a = _ACTIVE_SINK
    if a is None or a.tensor is None or not torch.is_grad_enabled() or not x.requires_grad:
        return None
    t = a.tensor
    if x.dtype != t.dtype or x.dim() != t.dim() or x.shape[1:] != t.shape[1:] or not x.is_contiguous() or x.device != t.device:
        return None
    row_bytes = t.stride(0) * t.element_size()
    off = x.data_ptr() - t.data_ptr()
    if off < 0 or off % row_bytes or off // row_bytes + x.shape[0] > t.shape[0]:
        return None
    return a


def _sink_add_dx(sink, x, dy2, w, w1=None):
    """sink[rows of x] += dy2 @ [w ; w1]  ([M, N] @ [N, Dv]); the first full-size contribution is taken as it comes.  True when done here."""
    M, Dv = dy2.shape[0], w.shape[1]
    N = dy2.shape[1]
    if sink.tensor is None or not gemm_f32s_nn_ok(M, Dv, N):      # (a second backward through a retained graph: the sink is spent -- ordinary gradients)
        return False
    r0, r1 = sink.rows(x)
    whole = r0 == 0 and r1 == sink.tensor.shape[0]
    if sink.buf is None and whole and w1 is None:
        sink.take(gemm_f32s_nn(dy2, w).view(sink.tensor.shape))
        return True
    buf = sink.full()
    gemm_f32s_nn_acc(dy2, w, buf[r0:r1].view(M, Dv), w1)
    return True


def _sink_add_dx_bf16(sink, x, dy2, wb):
    """bf16 storage mode: sink[rows of x] += dy2 [M,N] @ wb [N,K] -- the first full-size contribution is taken as it comes, later ones are added in the
    GEMM (beta = 1: one rounding of gradient + fp32 accumulator instead of a bf16 add kernel over the [B, T, D] tensor)."""
    M, K = dy2.shape[0], wb.shape[1]
    r0, r1 = sink.rows(x)
    if sink.buf is None and r0 == 0 and r1 == sink.tensor.shape[0]:
        sink.take(_mm_bf16_nn(dy2, wb).view(sink.tensor.shape))
    else:
        sink.full()[r0:r1].view(M, K).addmm_(dy2, wb)


def _sink_add(sink, x, g):
    """sink[rows of x] += g (a gradient that a kernel has already produced in full, e.g. K1g's dr): adopted when it is the first and covers the tensor."""
    r0, r1 = sink.rows(x)
    if sink.buf is None and r0 == 0 and r1 == sink.tensor.shape[0] and g.dtype == sink.tensor.dtype and g.is_contiguous():
        sink.take(g)
    else:
        sink.full()[r0:r1].add_(g)


def _dx_f32s(dy2: torch.Tensor, w_rows: torch.Tensor) -> torch.Tensor:
    """dX [M,K] = dY [M,N] @ W [N,K] in the f32s arithmetic: the own GEMM with the (small) weight transposed once where its tile
    constraints hold, else the generic split-precision product."""
    M, N = dy2.shape
    K = w_rows.shape[1]
    if gemm_f32s_ok(M, K, N):
        return gemm_f32s(dy2, transposed(w_rows))
    return _mm(dy2, w_rows, "f32s")


def _dw_f32s(dy2: torch.Tensor, x2: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
    """dW [N,K] = dY [M,N]^T @ X [M,K] in the f32s arithmetic (tsg_wgrad_f32s where its tiles fit, else fp32); ``out``: an [N,K] view
    (e.g. the video columns of a full-width parameter gradient) written in place."""
    M, N = dy2.shape
    K = x2.shape[1]
    if _WGRAD_KERNEL and wgrad_f32s_ok(M, N, K):
        if out is not None:
            return wgrad_f32s(dy2, x2, out=out)
        return wgrad_f32s(dy2, x2)[0]
    if out is not None:
        return out.copy_(dy2.t() @ x2)
    return dy2.t() @ x2


class _MatchHeadGemm(torch.autograd.Function):
    """K5 as the epilogue of its own first-Linear GEMM (tsg_match_head_gemm): x [B,T,K], w = W1[:, :K] (a column-slice VIEW of the
    [H, K + Kq] parameter, read in place), cs [B,H] = query half + bias, w2 [H], b2 [1] -> matching logits [B,T].  The pre-activation
    y = x w^T is written only when a gradient is needed (the backward kernel tsg_match_head_bwd reads it)."""

    @staticmethod
    @_fwd
    def forward(ctx, x, w, cs, w2, b2, act):
        require_device(x, w, cs, w2, b2)
        x = _f32c(x)
        B, T, K = x.shape
        H = w.shape[0]
        if w.stride(1) != 1 or w.stride(0) % 4 or w.data_ptr() % 16 or w.shape[1] != K:
            raise ValueError("match_head_gemm: w must be a row-major matrix or a column slice of one (16-byte aligned rows)")
        cs, w2, b2 = _f32p(cs), _f32p(w2).view(-1), _f32p(b2).view(-1)
        M = B * T
        need_y = any(ctx.needs_input_grad[:4])
        y = torch.empty(B, T, H, device=x.device, dtype=torch.float32) if need_y else None
        out = torch.empty(B, T, device=x.device, dtype=torch.float32)
        nb = int(load().tsg_head_gemm_ws_bytes(M, H, 1))
        ws = torch.empty(max(nb, 16), device=x.device, dtype=torch.uint8)
        _call("tsg_match_head_gemm", x, ptr(x), K, ptr(w), w.stride(0), ptr(cs), ptr(w2), ptr(b2), ptr(y) if need_y else None, ptr(out),
              ptr(ws), nb, M, T, H, K, int(act))
        if need_y:
            ctx.save_for_backward(x, w, y, cs, w2)
        ctx.act = int(act)
        return out

    @staticmethod
    @_bwd
    def backward(ctx, dl):
        x, w, y, cs, w2 = ctx.saved_tensors
        B, T, H = y.shape
        K = x.shape[2]
        dl = _f32p(dl)
        dy = torch.empty_like(y)
        dcs = torch.empty_like(cs); dw2 = torch.empty_like(w2); db2 = torch.empty(1, device=y.device, dtype=torch.float32)
        _call("tsg_match_head_bwd", y, ptr(y), ptr(cs), ptr(w2), ptr(dl), ptr(dy), ptr(dcs), ptr(dw2), ptr(db2), B, T, H, ctx.act, TSG_F32)
        dy2, x2 = dy.view(B * T, H), x.view(B * T, K)
        dx = _dx_f32s(dy2, w).view(B, T, K) if ctx.needs_input_grad[0] else None
        dw = _dw_f32s(dy2, x2) if ctx.needs_input_grad[1] else None
        return dx, dw, dcs, dw2, db2, None


def match_head_gemm(x, w, cs, w2, b2, activation="relu"):
    """Matching head fused into its first-Linear GEMM (include/tsg_hip.h: tsg_match_head_gemm)."""
    return _MatchHeadGemm.apply(x, w, cs, w2.reshape(-1), b2.reshape(-1), _ACTS[activation])


class _BoundaryHeadGemm(torch.autograd.Function):
    """K3 as the epilogue of its own first-Linear GEMM (tsg_boundary_head_gemm): x [B,T,K]; ws_, we_ = the video-half column slices of
    the start / end head's first Linear ([Hm, K] views, read in place as two row segments: no stacked copy); cs [B,2Hm], b1 / w2 [2Hm],
    b2 [2], gate [B,T] | None, mask [B,T] int | None -> (p_start, p_end) [B,T]."""

    @staticmethod
    @_fwd
    def forward(ctx, x, ws_, we_, cs, b1, w2, b2, gate, mask):
        require_device(x, ws_, we_, cs, b1, w2, b2, gate, mask)
        x = _f32c(x)
        B, T, K = x.shape
        Hm = ws_.shape[0]
        for w in (ws_, we_):
            if w.shape != (Hm, K) or w.stride(1) != 1 or w.stride(0) != ws_.stride(0) or w.stride(0) % 4 or w.data_ptr() % 16:
                raise ValueError("boundary_head_gemm: the two first-Linear slices must be [Hm,K] row-major views with one row stride")
        cs, b1, w2, b2 = _f32p(cs), _f32p(b1), _f32p(w2), _f32p(b2)
        gate_c = _f32p(gate) if gate is not None else None
        mask_c = mask.to(torch.int32).contiguous() if mask is not None else None
        need_y = any(ctx.needs_input_grad[:7]) or (gate is not None and ctx.needs_input_grad[7])
        J = 2 * Hm
        y = torch.empty(B, T, J, device=x.device, dtype=torch.float32) if need_y else None
        ps = torch.empty(B, T, device=x.device, dtype=torch.float32)
        pe = torch.empty_like(ps)
        nb = int(load().tsg_head_gemm_ws_bytes(B * T, J, 2))
        wsb = torch.empty(max(nb, 16), device=x.device, dtype=torch.uint8)
        _call("tsg_boundary_head_gemm", x, ptr(x), K, ptr(ws_), ptr(we_), ws_.stride(0), ptr(cs), ptr(b1), ptr(w2), ptr(b2),
              ptr(gate_c) if gate_c is not None else None, ptr(mask_c) if mask_c is not None else None, ptr(y) if need_y else None,
              ptr(ps), ptr(pe), ptr(wsb), nb, B, T, Hm, K)
        if need_y:
            ctx.save_for_backward(x, ws_, we_, y, cs, b1, w2, ps, pe, *([gate_c] if gate_c is not None else []),
                                  *([mask_c] if mask_c is not None else []))
        ctx.has_gate, ctx.has_mask = gate_c is not None, mask_c is not None
        return ps, pe

    @staticmethod
    @_bwd
    def backward(ctx, dps, dpe):
        saved = list(ctx.saved_tensors)
        x, ws_, we_, y, cs, b1, w2, ps, pe = saved[:9]
        rest = saved[9:]
        gate = rest.pop(0) if ctx.has_gate else None
        mask = rest.pop(0) if ctx.has_mask else None
        B, T, J = y.shape
        K, Hm = x.shape[2], J // 2
        dps = _f32p(dps) if dps is not None else torch.zeros_like(ps)
        dpe = _f32p(dpe) if dpe is not None else torch.zeros_like(pe)
        dy = torch.empty_like(y); dcs = torch.empty_like(cs)
        db1p = torch.empty(B, J, device=y.device, dtype=torch.float32)
        dw2p = torch.empty_like(db1p)
        db2p = torch.empty(B, 2, device=y.device, dtype=torch.float32)
        dgate = torch.empty(B, T, device=y.device, dtype=torch.float32) if gate is not None else None
        wk, nb = _k3_workspace(y.device, B, T, Hm)
        _k3_call("tsg_boundary_score_bwd_ws", y, ptr(y), ptr(cs), ptr(b1), ptr(w2),
              ptr(gate) if gate is not None else None, ptr(mask) if mask is not None else None,
              ptr(ps), ptr(pe), ptr(dps), ptr(dpe), ptr(dy), ptr(dcs), ptr(db1p), ptr(dw2p), ptr(db2p),
              ptr(dgate) if dgate is not None else None, ptr(wk), nb, B, T, Hm, TSG_F32)
        dy2, x2 = dy.view(B * T, J), x.view(B * T, K)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = _dx_f32s(dy2, torch.cat([ws_, we_], 0)).view(B, T, K)
        dws = dwe = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            dW = _dw_f32s(dy2, x2)                                     # [2Hm, K]: rows of the start head, then of the end head
            dws, dwe = dW[:Hm], dW[Hm:]
        return dx, dws, dwe, dcs, db1p.sum(0), dw2p.sum(0), db2p.sum(0), dgate, None


class _MatchHeadFull(torch.autograd.Function):
    """The matching head (K5) with the MODULE'S PARAMETERS as inputs: x [B,T,Dv], q [B,Dq], W1 [H, Dv+Dq], b1 [H], w2 [H], b2 [1] ->
    logits [B,T] (reference DistributionAlign.py:83-118, `Linear(cat(v, q)) -> act -> Linear(H,1)`).  The query half + bias is a
    per-item row (one small fp32 GEMM), the video half runs in tsg_match_head_gemm with W1[:, :Dv] read in place, and the backward
    returns ONE full-width dW1: the video columns written in place by the weight-gradient kernel, the query columns by one small
    GEMM -- no column-slice nodes in the graph (each cost a zero-filled [H, Dv+Dq] buffer, a copy and an accumulation per step)."""

    @staticmethod
    @_fwd
    def forward(ctx, x, q, W1, b1, w2, b2, act, sink=None):
        require_device(x, q, W1, b1, w2, b2)
        ctx.sink, ctx.x_in = sink, (x if sink is not None else None)
        x = _f32c(x)
        B, T, Dv = x.shape
        H = W1.shape[0]
        W1, b1, q = _f32p(W1), _f32p(b1), _f32p(q)
        ctx.shapes = (w2.shape, b2.shape)
        w2, b2 = _f32p(w2).view(-1), _f32p(b2).view(-1)
        cs = torch.addmm(b1, q, W1[:, Dv:].t())                        # [B,H]: query half + bias, plain fp32
        M = B * T
        need_y = any(ctx.needs_input_grad[:6])
        y = torch.empty(B, T, H, device=x.device, dtype=torch.float32) if need_y else None
        out = torch.empty(B, T, device=x.device, dtype=torch.float32)
        nb = int(load().tsg_head_gemm_ws_bytes(M, H, 1))
        ws = torch.empty(max(nb, 16), device=x.device, dtype=torch.uint8)
        _call("tsg_match_head_gemm", x, ptr(x), Dv, ptr(W1), W1.stride(0), ptr(cs), ptr(w2), ptr(b2), ptr(y) if need_y else None, ptr(out),
              ptr(ws), nb, M, T, H, Dv, int(act))
        if need_y:
            ctx.save_for_backward(x, q, W1, y, cs, w2)
        ctx.act = int(act)
        return out

    @staticmethod
    @_bwd
    def backward(ctx, dl):
        x, q, W1, y, cs, w2 = ctx.saved_tensors
        B, T, H = y.shape
        Dv = x.shape[2]
        dl = _f32p(dl)
        dy = torch.empty_like(y)
        dcs = torch.empty_like(cs); dw2 = torch.empty_like(w2); db2 = torch.empty(1, device=y.device, dtype=torch.float32)
        _call("tsg_match_head_bwd", y, ptr(y), ptr(cs), ptr(w2), ptr(dl), ptr(dy), ptr(dcs), ptr(dw2), ptr(db2), B, T, H, ctx.act, TSG_F32)
        dy2, x2 = dy.view(B * T, H), x.view(B * T, Dv)
        dx = None
        if ctx.needs_input_grad[0]:
            if ctx.sink is not None and _sink_add_dx(ctx.sink, ctx.x_in, dy2, W1[:, :Dv]):
                pass                                                     # summed in the shared activation's sink: None for autograd
            else:
                dx = _dx_f32s(dy2, W1[:, :Dv]).view(B, T, Dv)
        dq = dcs @ W1[:, Dv:] if ctx.needs_input_grad[1] else None
        dW1 = None
        if ctx.needs_input_grad[2]:
            dW1 = torch.empty_like(W1)
            _dw_f32s(dy2, x2, out=dW1[:, :Dv])
            dW1[:, Dv:] = dcs.t() @ q
        db1 = dcs.sum(0) if ctx.needs_input_grad[3] else None
        return dx, dq, dW1, db1, dw2.view(ctx.shapes[0]), db2.view(ctx.shapes[1]), None, None


def match_head_params(x, q, W1, b1, w2, b2, activation="relu"):
    """Matching head from its parameters (tsg_match_head_gemm; see _MatchHeadFull)."""
    return _MatchHeadFull.apply(x, q, W1, b1, w2, b2, _ACTS[activation], _sink_for(x))


class _BoundaryHeadFull(torch.autograd.Function):
    """The boundary head (K3) with the MODULE'S PARAMETERS as inputs: x [B,T,Dv], sent [B,Ds], the start / end heads' first Linears
    W [Hm, Dv+Ds], b [Hm] and second Linears w2 [1,Hm], b2 [1], gate [B,T] | None, mask [B,T] | None -> (p_start, p_end) [B,T]
    (reference SpanPredictor.py:71-85 on VideoSentenceConcat's rows).  As _MatchHeadFull: the video columns of the two first Linears are
    read in place by tsg_boundary_head_gemm, the small vectors are packed by one cat, and the backward returns full-width dW per head
    (video columns from the weight-gradient kernel in place, sentence columns from one small GEMM)."""

    @staticmethod
    @_fwd
    def forward(ctx, x, sent, Ws, bs, We, be, w2s, b2s, w2e, b2e, gate, mask, sink=None):
        require_device(x, sent, Ws, bs, We, be, w2s, b2s, w2e, b2e, gate, mask)
        ctx.sink, ctx.x_in = sink, (x if sink is not None else None)
        x = _f32c(x)
        B, T, Dv = x.shape
        Hm, J = Ws.shape[0], 2 * Ws.shape[0]
        Ws, We, sent = _f32p(Ws), _f32p(We), _f32p(sent)
        if We.shape != Ws.shape or Ws.stride(0) != We.stride(0) or Ws.stride(0) % 4 or Ws.data_ptr() % 16 or We.data_ptr() % 16 or Dv % 4:
            raise ValueError("boundary_head: the two first Linears must be [Hm, Dv+Ds] row-major parameters of one shape")
        cs = torch.cat([sent @ Ws[:, Dv:].t(), sent @ We[:, Dv:].t()], 1)              # [B,2Hm]: sentence half (b1 is added in the kernel)
        pack = torch.cat([t.reshape(-1).to(torch.float32) for t in (bs, be, w2s, w2e, b2s, b2e)])
        b1, w2, b2 = pack[:J], pack[J:2 * J], pack[2 * J:2 * J + 2]
        gate_c = _f32p(gate) if gate is not None else None
        mask_c = mask.to(torch.int32).contiguous() if mask is not None else None
        need_y = any(ctx.needs_input_grad[:11])
        y = torch.empty(B, T, J, device=x.device, dtype=torch.float32) if need_y else None
        ps = torch.empty(B, T, device=x.device, dtype=torch.float32)
        pe = torch.empty_like(ps)
        nb = int(load().tsg_head_gemm_ws_bytes(B * T, J, 2))
        wsb = torch.empty(max(nb, 16), device=x.device, dtype=torch.uint8)
        _call("tsg_boundary_head_gemm", x, ptr(x), Dv, ptr(Ws), ptr(We), Ws.stride(0), ptr(cs), ptr(b1), ptr(w2), ptr(b2),
              ptr(gate_c) if gate_c is not None else None, ptr(mask_c) if mask_c is not None else None, ptr(y) if need_y else None,
              ptr(ps), ptr(pe), ptr(wsb), nb, B, T, Hm, Dv)
        if need_y:
            ctx.save_for_backward(x, sent, Ws, We, y, cs, b1, w2, ps, pe, *([gate_c] if gate_c is not None else []),
                                  *([mask_c] if mask_c is not None else []))
        ctx.has_gate, ctx.has_mask = gate_c is not None, mask_c is not None
        ctx.shapes = (w2s.shape, w2e.shape)
        return ps, pe

    @staticmethod
    @_bwd
    def backward(ctx, dps, dpe):
        saved = list(ctx.saved_tensors)
        x, sent, Ws, We, y, cs, b1, w2, ps, pe = saved[:10]
        rest = saved[10:]
        gate = rest.pop(0) if ctx.has_gate else None
        mask = rest.pop(0) if ctx.has_mask else None
        B, T, J = y.shape
        Dv, Hm = x.shape[2], J // 2
        dps = _f32p(dps) if dps is not None else torch.zeros_like(ps)
        dpe = _f32p(dpe) if dpe is not None else torch.zeros_like(pe)
        dy = torch.empty_like(y); dcs = torch.empty_like(cs)
        db1p = torch.empty(B, J, device=y.device, dtype=torch.float32)
        dw2p = torch.empty_like(db1p)
        db2p = torch.empty(B, 2, device=y.device, dtype=torch.float32)
        dgate = torch.empty(B, T, device=y.device, dtype=torch.float32) if gate is not None else None
        wk, nb = _k3_workspace(y.device, B, T, Hm)
        _k3_call("tsg_boundary_score_bwd_ws", y, ptr(y), ptr(cs), ptr(b1), ptr(w2),
              ptr(gate) if gate is not None else None, ptr(mask) if mask is not None else None,
              ptr(ps), ptr(pe), ptr(dps), ptr(dpe), ptr(dy), ptr(dcs), ptr(db1p), ptr(dw2p), ptr(db2p),
              ptr(dgate) if dgate is not None else None, ptr(wk), nb, B, T, Hm, TSG_F32)
        dy2, x2 = dy.view(B * T, J), x.view(B * T, Dv)
        ws_, we_ = Ws[:, :Dv], We[:, :Dv]
        dx = None
        if ctx.needs_input_grad[0]:
            if ctx.sink is not None and _sink_add_dx(ctx.sink, ctx.x_in, dy2, ws_, we_):
                pass                                                     # added onto the shared activation's gradient rows: None for autograd
            else:
                dx = _dx_f32s(dy2, torch.cat([ws_, we_], 0)).view(B, T, Dv)
        dsent = torch.addmm(dcs[:, :Hm] @ Ws[:, Dv:], dcs[:, Hm:], We[:, Dv:]) if ctx.needs_input_grad[1] else None
        dWs = dWe = None
        if ctx.needs_input_grad[2] or ctx.needs_input_grad[4]:
            dW = torch.empty(2, Hm, Ws.shape[1], device=y.device, dtype=torch.float32)       # [start ; end], full width
            dW2 = dW.view(J, Ws.shape[1])
            _dw_f32s(dy2, x2, out=dW2[:, :Dv])
            dW2[:, Dv:] = dcs.t() @ sent
            dWs, dWe = dW[0], dW[1]
        db1, dw2, db2 = db1p.sum(0), dw2p.sum(0), db2p.sum(0)
        s2s, s2e = ctx.shapes
        return (dx, dsent, dWs, db1[:Hm], dWe, db1[Hm:], dw2[:Hm].view(s2s), db2[0:1], dw2[Hm:].view(s2e), db2[1:2], dgate, None, None)


def boundary_head_params(x, sent, Ws, bs, We, be, w2s, b2s, w2e, b2e, gate=None, mask=None):
    """Boundary head from its parameters (tsg_boundary_head_gemm; see _BoundaryHeadFull)."""
    return _BoundaryHeadFull.apply(x, sent, Ws, bs, We, be, w2s, b2s, w2e, b2e, gate, mask, _sink_for(x))


def boundary_head_gemm(x, w_start, w_end, cs, b1, w2, b2, gate=None, mask=None):
    """Boundary head fused into its first-Linear GEMM (include/tsg_hip.h: tsg_boundary_head_gemm)."""
    return _BoundaryHeadGemm.apply(x, w_start, w_end, cs, b1, w2, b2, gate, mask)


class _LayerNorm(torch.autograd.Function):
    """LayerNorm over the last axis (tsg_layer_norm_fwd / _bwd): x [..., d] fp32 (or bf16 in the storage mode), gamma / beta [d] fp32."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        require_device(x, gamma, beta)
        bf = x.dtype == _BF
        x = _act(x, bf)
        gamma, beta = _f32p(gamma), _f32p(beta)
        d = x.shape[-1]
        rows = x.numel() // d
        y = torch.empty_like(x)
        mean = torch.empty(rows, device=x.device, dtype=torch.float32)
        rstd = torch.empty_like(mean)
        ctx.dt = TSG_BF16 if bf else TSG_F32
        _call("tsg_layer_norm_fwd", x, ptr(x), ptr(gamma), ptr(beta), ptr(y), ptr(mean), ptr(rstd), rows, d, float(eps), ctx.dt)
        ctx.save_for_backward(x, gamma, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, mean, rstd = ctx.saved_tensors
        dy = _act(dy, ctx.dt == TSG_BF16)
        d = x.shape[-1]
        rows = x.numel() // d
        dx = torch.empty_like(x)
        dgamma = torch.empty_like(gamma); dbeta = torch.empty_like(gamma)
        nb = int(load().tsg_layer_norm_bwd_ws_bytes(rows, d))
        ws = torch.empty(nb, device=x.device, dtype=torch.uint8)
        _call("tsg_layer_norm_bwd", x, ptr(x), ptr(dy), ptr(gamma), ptr(mean), ptr(rstd), ptr(dx), ptr(dgamma), ptr(dbeta), ptr(ws), nb, rows, d, ctx.dt)
        return dx, dgamma, dbeta, None


def layer_norm_ok(x: torch.Tensor) -> bool:
    # (not under torch.autocast: F.layer_norm autocasts to fp32 in and out there, and this Function is not a custom_fwd one -- ADVICE r4)
    return not torch.is_autocast_enabled() and x.is_cuda and x.dtype in (torch.float32, _BF) and x.shape[-1] % 4 == 0 and x.shape[-1] <= 2048 and os.environ.get("TSG_LN", "1") != "0"


def layer_norm(x, gamma, beta, eps=1e-5):
    """nn.LayerNorm(d) on the hand-written kernels (include/tsg_hip.h: tsg_layer_norm_fwd / _bwd)."""
    return _LayerNorm.apply(x, gamma, beta, eps)


_OWN_DROPOUT = os.environ.get("TSG_DROPOUT", "own") != "torch"          # A/B switch: "torch" = F.dropout (byte mask written and re-read)


_graph_keys = {}       # device -> [persistent int32 [kKeySlots, 2] buffer, next slot]
_graph_key_buffers = []   # every key buffer ever created: never freed


def _graph_key_slot(device):
    """Two key words for ONE captured dropout call, in memory that does not belong to the graph's pool.  Round 5: as ``torch.empty(2)`` inside the
    capture the keys of the three dropouts of a GMD step shared their 512-byte pool blocks with a gradient allocated later in the same graph
    (``word_embed.bias.grad``, 1200 bytes = the three blocks), and in back-to-back replays of the two graphs of ``GraphedTrainStep`` the key words
    showed up INSIDE that gradient -- once in ~100 replays as a NaN pattern, which then stuck in the parameter (bench.py --dtype bf16 --steps 400:
    non-finite; with a host synchronisation between the graphs, or with keys that do not share blocks, 400 steps stay finite).  The slots are
    created outside the capture (``mha_graph_rng``), each captured call takes the next one, a slot is never reused by another call site of the
    same graph."""
    dev = _cuda_device(device)
    st = _graph_keys.get(dev)
    if st is None:
        raise RuntimeError("dropout under graph capture needs functional.mha_graph_rng(device) called before the capture (engine.GraphedTrainStep does)")
    buf, nxt = st
    if nxt >= buf.shape[0]:
        raise RuntimeError(f"more than {buf.shape[0]} dropout calls captured on {dev}: raise functional.kKeySlots")
    st[1] = nxt + 1
    return buf[nxt]


kKeySlots = 4096


class _Dropout(torch.autograd.Function):
    """Dropout without a stored mask (tsg_dropout): the backward is the same launch on the gradient, the mask regenerated from the keys."""

    @staticmethod
    def forward(ctx, x, p, seed, offset, rng):
        require_device(x)
        x = x.contiguous()
        y = torch.empty_like(x)
        ctx.dt = TSG_BF16 if x.dtype == _BF else TSG_F32
        ctx.p, ctx.seed, ctx.offset = p, seed, offset
        if rng is not None:                                          # under graph capture: (seed, offset) in device memory, keys kept for the backward
            keys = _graph_key_slot(x.device)
            _call("tsg_dropout", x, ptr(x), ptr(y), x.numel(), p, 0, 0, ptr(rng), ptr(keys), 1, ctx.dt)
            ctx.save_for_backward(keys)
        else:
            _call("tsg_dropout", x, ptr(x), ptr(y), x.numel(), p, seed, offset, None, None, 0, ctx.dt)
        ctx.has_keys = rng is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        if ctx.has_keys:
            _call("tsg_dropout", dy, ptr(dy), ptr(dx), dy.numel(), ctx.p, 0, 0, None, ptr(ctx.saved_tensors[0]), 2, ctx.dt)
        else:
            _call("tsg_dropout", dy, ptr(dy), ptr(dx), dy.numel(), ctx.p, ctx.seed, ctx.offset, None, None, 0, ctx.dt)
        return dx, None, None, None, None


def dropout_ok(x) -> bool:
    return not torch.is_autocast_enabled() and _OWN_DROPOUT and x.is_cuda and x.dtype in (torch.float32, _BF) and x.numel() > 0


def dropout(x, p, training=True):
    """F.dropout(x, p, training) on the hand-written kernel: the keep decision of element i is a counter-based hash of i and of
    (seed, offset) -- seed = torch.initial_seed(), offset drawn from torch's CPU generator per call, so ``torch.manual_seed`` makes it
    reproducible and every call draws a fresh mask; under a HIP-graph capture the pair lives in device memory and the captured
    increment of the offset makes every REPLAY draw a fresh mask (the state of ``mha_graph_rng``)."""
    if not training or p <= 0.0:
        return x
    if p >= 1.0:
        return x * 0.0
    seed, offset, rng = torch.initial_seed() & 0xFFFFFFFFFFFFFFFF, 0, None
    if torch.distributed.is_available() and torch.distributed.is_initialized():      # ranks seeded alike still draw different masks
        seed = (seed + 0x9E3779B97F4A7C15 * torch.distributed.get_rank()) & 0xFFFFFFFFFFFFFFFF
    if torch.cuda.is_current_stream_capturing():
        rng = _mha_rng_state.get(_cuda_device(x.device))
        if rng is None:
            raise RuntimeError("dropout under graph capture needs functional.mha_graph_rng(device) called before the capture "
                               "(engine.GraphedTrainStep does)")
        rng[1] += 1                                                  # captured: advances on every replay
    else:
        offset = int(torch.randint(0, 2 ** 62, (1,)).item())
    return _Dropout.apply(x, float(p), seed, offset, rng)


class _MomentPool(torch.autograd.Function):
    """MomentPooling's three masked means in one pass (tsg_moment_pool_fwd / _bwd): feat [B,T,D] (fp32, or bf16 in the storage
    mode), masks float [B,T] x 3 -> (target, fore, back) means, fp32 [B,D] each (views of one [B,3,D] buffer)."""

    @staticmethod
    def forward(ctx, feat, m_target, m_fore, m_back, sink=None):
        require_device(feat, m_target, m_fore, m_back)
        ctx.sink, ctx.x_in = sink, (feat if sink is not None else None)
        bf = feat.dtype == _BF
        feat = _act(feat, bf)
        ms = tuple(_f32p(m) for m in (m_target, m_fore, m_back))
        B, T, D = feat.shape
        if any(m.shape != (B, T) for m in ms):
            raise ValueError(f"moment_pool: masks must be [{B},{T}]")
        pooled = torch.empty(B, 3, D, device=feat.device, dtype=torch.float32)
        ctx.dt = TSG_BF16 if bf else TSG_F32
        _call("tsg_moment_pool_fwd", feat, ptr(feat), ptr(ms[0]), ptr(ms[1]), ptr(ms[2]), ptr(pooled), B, T, D, ctx.dt)
        ctx.save_for_backward(*ms)
        ctx.shape = (B, T, D)
        return pooled[:, 0], pooled[:, 1], pooled[:, 2]      # three outputs: their gradients arrive as three [B,D] tensors (as one [B,3,D]
                                                             # selected three times, autograd zero-filled and accumulated three [B,3,D] buffers)

    @staticmethod
    def backward(ctx, dt, df, db):
        ms = ctx.saved_tensors
        B, T, D = ctx.shape
        like = next(g for g in (dt, df, db) if g is not None)
        dpooled = torch.stack([g if g is not None else torch.zeros_like(like) for g in (dt, df, db)], 1)
        dpooled = _f32p(dpooled)
        dfeat = torch.empty(B, T, D, device=dpooled.device, dtype=_BF if ctx.dt == TSG_BF16 else torch.float32)
        _call("tsg_moment_pool_bwd", dpooled, ptr(dpooled), ptr(ms[0]), ptr(ms[1]), ptr(ms[2]), ptr(dfeat), B, T, D, ctx.dt)
        sink = ctx.sink
        if sink is not None and sink.tensor is not None and dfeat.dtype == sink.tensor.dtype:
            r0, r1 = sink.rows(ctx.x_in)
            if sink.buf is None and r0 == 0 and r1 == sink.tensor.shape[0]:
                sink.take(dfeat)                                         # the first consumer to run (it is created last): the sink adopts its gradient
            else:
                sink.full()[r0:r1].add_(dfeat)
            return None, None, None, None, None
        return dfeat, None, None, None, None


def moment_pool(feat, m_target, m_fore, m_back):
    """-> (target, fore, back) [B,D]: masked means of feat over the three ranges (include/tsg_hip.h: tsg_moment_pool_fwd)."""
    return _MomentPool.apply(feat, m_target, m_fore, m_back, _sink_for(feat))


class _GmdLosses(torch.autograd.Function):
    """The four GMD training losses (K4, csrc/losses.hip) -> (parts[4] = span, matching BCE, matching KL, order CE, all
    un-weighted; total = span + lam[0] BCE + lam[1] KL + lam[2] CE)."""

    @staticmethod
    @_fwd
    def forward(ctx, ps, pe, om, pm, od, pd, fs, pfs, tl, ptl, vm, lam):
        require_device(ps, pe, om, pm, od, pd, fs, pfs, tl, ptl, vm)
        ps, pe, om, pm, od, pd = (_f32c(t) for t in (ps, pe, om, pm, od, pd))
        tl, ptl, vm = (t.to(torch.float32).contiguous() for t in (tl, ptl, vm))     # labels / masks arrive as int or bool
        fs, pfs = fs.to(torch.long).contiguous(), pfs.to(torch.long).contiguous()
        B, T = om.shape
        for t in (ps, pe, pm, tl, ptl, vm):
            if t.shape != (B, T):
                raise ValueError(f"gmd_losses: expected [{B},{T}] tensors, got {tuple(t.shape)}")
        if od.shape != (B, 2) or pd.shape != (B, 2) or fs.shape != (B, 2) or pfs.shape != (B, 2):
            raise ValueError("gmd_losses: od / pd / fs / pfs must be [B,2]")
        ws = torch.empty(8, device=om.device, dtype=torch.float32)
        out = torch.empty(5, device=om.device, dtype=torch.float32)
        lam = tuple(float(v) for v in lam)
        _call("tsg_gmd_losses_fwd", om, ptr(ps), ptr(pe), ptr(om), ptr(pm), ptr(od), ptr(pd), ptr(fs), ptr(pfs), ptr(tl), ptr(ptl),
              ptr(vm), ptr(ws), ptr(out), B, T, *lam)
        ctx.save_for_backward(ps, pe, om, pm, od, pd, fs, pfs, tl, ptl, vm, ws)
        ctx.lam = lam
        ctx.set_materialize_grads(False)
        return out[:4], out[4]

    @staticmethod
    @_bwd
    def backward(ctx, dL, dtot):
        ps, pe, om, pm, od, pd, fs, pfs, tl, ptl, vm, ws = ctx.saved_tensors
        B, T = om.shape
        dps, dpe, dom, dpm = (torch.empty_like(om) for _ in range(4))
        dod, dpd = torch.empty_like(od), torch.empty_like(pd)
        if dL is None and dtot is None:
            dL = torch.zeros(4, device=om.device, dtype=torch.float32)
        dL = _f32c(dL) if dL is not None else None
        dtot = _f32c(dtot).reshape(1) if dtot is not None else None
        _call("tsg_gmd_losses_bwd", om, ptr(ps), ptr(pe), ptr(om), ptr(pm), ptr(od), ptr(pd), ptr(fs), ptr(pfs), ptr(tl), ptr(ptl),
              ptr(vm), ptr(ws), ptr(dL) if dL is not None else None, ptr(dtot) if dtot is not None else None,
              ptr(dps), ptr(dpe), ptr(dom), ptr(dpm), ptr(dod), ptr(dpd), B, T, *ctx.lam)
        return dps, dpe, dom, dpm, dod, dpd, None, None, None, None, None, None


def gmd_losses(ps, pe, om, pm, od, pd, fs, pfs, tl, ptl, vm, lam=(1.0, 1.0, 1.0)):
    """-> (parts [4]: span_ground_loss, BCE(om)+BCE(pm), matching KL, order-discrimination CE -- un-weighted;
    total = parts[0] + lam[0] parts[1] + lam[1] parts[2] + lam[2] parts[3])   (include/tsg_hip.h, K4)."""
    return _GmdLosses.apply(ps, pe, om, pm, od, pd, fs, pfs, tl, ptl, vm, lam)


_WGRAD_KERNEL = os.environ.get("TSG_WGRAD", "1") != "0"      # A/B switch: 0 = library GEMMs for the weight gradients


_ONES = {}


def _colsum(t2: torch.Tensor) -> torch.Tensor:
    """Column sums of an fp32 [M,N] matrix (a bias gradient): on the GPU as a one-row fp32 GEMM ones[1,M] @ t2 -- torch's dim-0 reduction
    takes 20 us for [2560 x 1024] (one wave per column block), the GEMM 5."""
    M = t2.shape[0]
    if not (t2.is_cuda and t2.dtype in (torch.float32, _BF) and M >= 256 and t2.dim() == 2):
        return t2.sum(0, dtype=torch.float32)
    key = (t2.device, M, t2.dtype)
    ones = _ONES.get(key)
    if ones is None:
        if torch.cuda.is_current_stream_capturing():
            return t2.sum(0, dtype=torch.float32)
        ones = _ONES[key] = torch.ones(1, M, device=t2.device, dtype=t2.dtype)
    if t2.dtype == _BF:
        return torch.mm(ones, t2, out_dtype=torch.float32).view(-1)     # bf16 rows, fp32 sums (the bias gradient is never rounded to bf16)
    return torch.mm(ones, t2).view(-1)


class _LinearSplit(torch.autograd.Function):
    """y = x w^T (+ b) with the two large GEMMs (forward, input gradient) in the split-precision mode on the library's bf16
    kernels; the weight gradient dY^T x has a small output and a T*B-long contraction over the ROWS of both operands, a shape
    the library is slow at in bf16 (and 16x below the bf16 peak in fp32): tsg_wgrad_f32s (csrc/wgrad_split.hip) where its
    tile constraints hold, the fp32 GEMM otherwise."""

    @staticmethod
    @_fwd
    def forward(ctx, x, w, b):
        x2 = _f32c(x).view(-1, x.shape[-1])
        ctx.mode = _GEMM_DTYPE                                                # the backward runs in the forward's mode, whatever is current then
        if _GEMM_DTYPE == "f32s" and x2.is_cuda and w.is_contiguous() and gemm_f32s_ok(x2.shape[0], w.shape[0], w.shape[1]):
            y = gemm_f32s(x2, w, b)                                           # bias in the GEMM's epilogue
        else:
            y = _mm(x2, w.t())
            if b is not None:
                y += b
        ctx.save_for_backward(x2, w)
        ctx.has_bias = b is not None
        return y.view(*x.shape[:-1], w.shape[0])

    @staticmethod
    @_bwd
    def backward(ctx, dy):
        x2, w = ctx.saved_tensors
        dy2 = _f32c(dy).view(-1, w.shape[0])
        mode = ctx.mode
        dx = _mm(dy2, w, mode).view(*dy.shape[:-1], w.shape[1]) if ctx.needs_input_grad[0] else None
        dw = None
        if ctx.needs_input_grad[1]:
            M, (N, K) = x2.shape[0], w.shape
            if mode == "f32s" and _WGRAD_KERNEL and wgrad_f32s_ok(M, N, K):
                # dY^T X in the same split-precision arithmetic by the hand-written kernel that converts the fp32 rows on load:
                # 132 us at [1024 x 16384] x [16384 x 1024] against 290 us for the fp32 library GEMM and 340 us for operand planes
                # + the library's bf16 GEMM (tools/wgrad_time.py)
                dw = wgrad_f32s(dy2, x2)[0]
            elif mode == "f32s" and N * K >= 2048 * 2048 and M % 16 == 0 and N % 4 == 0 and K % 4 == 0:
                # a LARGE weight gradient (the 2048 x 2048 projections of the self-attention head: 69 GFLOP each, 0.49 ms as an
                # fp32 GEMM) as a split-precision GEMM over the row contraction: both operands K-contiguous from the transposing
                # split, as in the LSTM weight gradients
                At = torch.empty(N, 3 * M, device=x2.device, dtype=torch.bfloat16)
                Bt = torch.empty(K, 3 * M, device=x2.device, dtype=torch.bfloat16)
                split_bf16x3_t(dy2, 0, N, 0, False, At)
                split_bf16x3_t(x2, 0, K, 0, True, Bt)
                dw = torch.mm(At, Bt.t(), out_dtype=torch.float32)
            else:
                dw = dy2.t() @ x2
        db = _colsum(dy2) if ctx.has_bias and ctx.needs_input_grad[2] else None
        return dx, dw, db


_OWN_GEMM_BF16 = os.environ.get("TSG_OWN_GEMM_BF16", "1") != "0"     # 0: torch.mm -> hipBLASLt for the bf16 mode's projections (A/B)


_OWN_GEMM_BF16_ALL = os.environ.get("TSG_OWN_GEMM_BF16") == "all"   # every shape the kernel takes, also where the library is faster (A/B)


def gemm_bf16_ok(M: int, N: int, K: int) -> bool:
    """Shapes tsg_gemm_bf16 takes (include/tsg_hip.h: whole 128 / 256-row x 256-column tiles, 32-deep K chunks) AND is the faster path for
    (profiles/r5/gemm_bf16_vs_library_v1.txt, one MI355X, us own / hipBLASLt NT): up to one tile per CU -- [2560 x 1024] . [1024 x 1024]^T 19.7 / 24.9
    (the library's NT pick at that shape: 263), [1280 x 1024] . [4096 x 1024]^T 22.6 / 25.8, [8192 x 1024] . [1024 x 1024]^T 25.1 / 26.2,
    [16384 x 2048] . [512 x 2048]^T 37.2 / 40.8.  With several tiles per CU the kernel is LDS-bound (24 fragment reads + the DMA's writes per
    32 MFMAs: the LDS pipe is as busy as the matrix pipe; ablations in the same file) and the library's 4-wave 128 x 128 wave tiles win:
    [16384 x 1024] . [4096 x 1024]^T 158 / 122 -- those products stay on the library.  TSG_OWN_GEMM_BF16=all / 0: every shape / none."""
    if not (_OWN_GEMM_BF16 and M > 0 and M % 128 == 0 and N % 256 == 0 and K % 32 == 0):
        return False
    tm = 256 if M % 256 == 0 else 128
    return _OWN_GEMM_BF16_ALL or (M // tm) * (N // 256) < 256


def gemm_bf16(x: torch.Tensor, w: torch.Tensor, bias: torch.Tensor = None, out_dtype=None) -> torch.Tensor:
    """y [M,N] = x [M,K] @ w [N,K]^T (+ bias) on the hand-written bf16 GEMM (csrc/gemm_bf16.hip: operands by LDS-DMA, fp32 accumulate);
    x, w bf16 with contiguous rows (column slices of wider matrices go in as they are), bias fp32, y bf16 (default) or fp32."""
    require_device(x, w, bias)
    out_dtype = _BF if out_dtype is None else out_dtype
    M, K = x.shape
    N = w.shape[0]
    if x.dtype != _BF or w.dtype != _BF or x.stride(1) != 1 or w.stride(1) != 1 or w.shape[1] != K:
        raise ValueError(f"gemm_bf16: needs bf16 operands with contiguous rows, x{tuple(x.shape)} w{tuple(w.shape)}")
    y = torch.empty(M, N, device=x.device, dtype=out_dtype)
    _call("tsg_gemm_bf16", x, ptr(x), x.stride(0), ptr(w), w.stride(0), ptr(_f32c(bias)) if bias is not None else None, ptr(y), N, M, N, K,
          TSG_BF16 if out_dtype == _BF else TSG_F32)
    return y


def _mm_bf16_nt(x2: torch.Tensor, wb: torch.Tensor, bias=None) -> torch.Tensor:
    """x2 [M,K] bf16 @ wb [N,K]^T (+ bias) -> bf16: the own kernel where its tiling applies, else the library."""
    if x2.is_cuda and gemm_bf16_ok(x2.shape[0], wb.shape[0], x2.shape[1]) and x2.stride(1) == 1 and x2.stride(0) % 8 == 0 and x2.data_ptr() % 16 == 0:
        return gemm_bf16(x2, wb, bias)
    return torch.mm(x2, wb.t()) if bias is None else torch.addmm(bias.detach().to(_BF), x2, wb.t())


def _mm_bf16_nn(dy2: torch.Tensor, wb: torch.Tensor) -> torch.Tensor:
    """dy2 [M,N] bf16 @ wb [N,K] -> bf16 (the input gradient of a Linear): the own kernel on a transposed bf16 copy of the weight (8 MB for an
    LSTM layer's W_ih: one small pass against a 100-us-class product), else the library's NN form."""
    if dy2.is_cuda and gemm_bf16_ok(dy2.shape[0], wb.shape[1], dy2.shape[1]) and dy2.stride(1) == 1 and dy2.stride(0) % 8 == 0 and dy2.data_ptr() % 16 == 0:
        return gemm_bf16(dy2, wb.t().contiguous())
    return torch.mm(dy2, wb)


class _LinearBf16(torch.autograd.Function):
    """y = x w^T (+ b) in the bf16 storage mode: bf16 activations in and out, the fp32 parameter rounded to bf16 once per use
    (kept for the backward), plain bf16 MFMA GEMMs with fp32 accumulation; the weight / bias gradients come out of their GEMM /
    reduction in fp32 (the master gradient is never rounded to bf16)."""

    @staticmethod
    def forward(ctx, x, w, b, sink=None):
        ctx.sink, ctx.x_in = sink, (x if sink is not None else None)
        x2 = _bfc(x).view(-1, x.shape[-1])
        wb = weight_bf16(w)
        y = _mm_bf16_nt(x2, wb, b)
        ctx.save_for_backward(x2, wb)
        ctx.has_bias, ctx.xshape = b is not None, x.shape
        return y.view(*x.shape[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, wb = ctx.saved_tensors
        dy2 = _bfc(dy).view(-1, wb.shape[0])
        dx = None
        if ctx.needs_input_grad[0]:
            if ctx.sink is not None and ctx.sink.tensor is not None:
                _sink_add_dx_bf16(ctx.sink, ctx.x_in, dy2, wb)          # summed in the shared activation's sink: None for autograd
            else:
                dx = _mm_bf16_nn(dy2, wb).view(ctx.xshape)
        dw = None
        if ctx.needs_input_grad[1]:
            M, (N, K) = x2.shape[0], wb.shape
            if _WGRAD_KERNEL and wgrad_f32s_ok(M, N, K):
                dw = wgrad_bf16(dy2, x2)[0]                    # tsg_wgrad_bf16 (csrc/wgrad_split.hip): dY^T X on the hand-written kernel
            else:
                dw = torch.mm(dy2.t(), x2, out_dtype=torch.float32)
        db = _colsum(dy2) if ctx.has_bias and ctx.needs_input_grad[2] else None
        return dx, dw, db, None


def linear(x, w, b=None):
    """torch.nn.functional.linear; in the "f32s" GEMM mode the large projections of the path (>= 2048 rows: SCDM W_a
    attention.py:104-106, the matching head's and the boundary head's first Linear) run as split-precision GEMMs; in the bf16
    storage mode every projection of the path is a bf16 GEMM with bf16 output (``_LinearBf16``)."""
    if bf16_storage() and x.is_cuda:
        return _LinearBf16.apply(x, w, b, _sink_for(x) if x.dtype == _BF else None)
    if x.dtype == _BF and w.dtype != _BF:                    # a bf16 activation met outside the storage mode's context
        return torch.nn.functional.linear(x.float(), w, b)
    rows = x.numel() // max(x.shape[-1], 1)
    if _GEMM_DTYPE == "f32s" and x.is_cuda and rows >= 2048 and rows % 4 == 0 and x.shape[-1] % 4 == 0 and w.shape[0] % 4 == 0:
        return _LinearSplit.apply(x, w, b)
    return torch.nn.functional.linear(x, w, b)


def linear_hip(x, w, b=None):
    """torch.nn.functional.linear on the hand-written fp32 MFMA GEMM (include/tsg_hip.h: tsg_linear_fwd).  Opt-in: rocBLAS
    is ~18 % faster at the path's shapes (HISTORY.md, old section 4 "Projection GEMMs"), so the modules keep F.linear."""
    return _LinearHip.apply(x, w, b)


def _lstm_fwd_ws(B, T, h, device):
    """Workspace of ``tsg_lstm_fwd_ws`` (int32 tensor, byte count): TSG_LSTM_SYNC_BYTES of sync words, followed -- for the hidden sizes
    the persistent kernel takes -- by its exchange ring.  A fresh allocation per call (the caching allocator returns the same block in a
    steady-state step; inside a graph capture it belongs to the graph's pool), never shared between launches that may overlap."""
    nb = max(int(load().tsg_lstm_fwd_ws_bytes(B, T, h)), 2048)
    return torch.empty(nb // 4, device=device, dtype=torch.int32), nb


class _BiLSTMLayer(torch.autograd.Function):
    """One bidirectional LSTM layer from zero state.  x [T,B,I] (time-major) or, with ``bm``, [B,T,I] (batch-major, the
    model's layout: the kernels index the sequence tensors either way, so no transposed copies surround the recurrence);
    W_ih [8h,I] (forward rows, then reverse), bias [8h] (b_ih + b_hh), W_hh [2,4h,h]  ->  out [T,B,2h] / [B,T,2h] and the
    cell states Cs [T,2,B,h] (always time-major).  The input GEMM and the weight-gradient GEMMs are library GEMMs (rocBLAS
    via torch); the recurrence is libtsg_hip.so."""

    @staticmethod
    @_fwd
    def forward(ctx, x, W_ih, bias, W_hh, bm=False, mode=_AUTO, bias2=None):
        require_device(x, W_ih, bias, W_hh, bias2)
        mode = ctx.mode = _GEMM_DTYPE if mode is _AUTO else mode      # kept for the backward (it used to read the global)
        x, W_ih, bias, W_hh = _f32c(x), _f32c(W_ih), _f32c(bias), _f32c(W_hh)
        ctx.two_biases = bias2 is not None
        own = mode == "f32s" and _LSTM_OWN_GEMM and x.is_cuda and bias2 is not None and bias2.numel() == bias.numel()
        if bias2 is not None and not (own and gemm_f32s_ok(x.numel() // x.shape[-1], W_ih.shape[0], x.shape[-1])):
            bias, bias2 = bias + _f32c(bias2), None                   # (elsewhere the two nn.LSTM biases are added up front, as before)
        (B, T, I) = x.shape if bm else (x.shape[1], x.shape[0], x.shape[2])
        h = W_hh.shape[2]
        if W_ih.shape != (8 * h, I) or W_hh.shape != (2, 4 * h, h) or bias.numel() != 8 * h:
            raise ValueError(f"bilstm: shape mismatch x{tuple(x.shape)} W_ih{tuple(W_ih.shape)} W_hh{tuple(W_hh.shape)}")
        ctx.Ws = None
        ctx.own = False
        if mode is None:
            Gx, kbias = torch.addmm(bias, x.view(T * B, I), W_ih.t()), None   # [rows,2,4h]; bias in the GEMM epilogue
        elif mode == "f32s" and _LSTM_OWN_GEMM and x.is_cuda and gemm_f32s_ok(T * B, 8 * h, I):
            # the LSTM's matrix work on the hand-written kernels, operands converted on load: no operand planes in HBM at all
            # (forward: tsg_gemm_f32s; backward: tsg_gemm_f32s for dX, tsg_wgrad_f32s for [dW_ih | dW_hh] of both directions)
            ctx.own = True
            # two biases (nn.LSTM's b_ih, b_hh): one rides in the GEMM's epilogue, the other is added by the recurrence kernel -- no add kernel
            Gx, kbias = gemm_f32s(x.view(T * B, I), W_ih, _f32c(bias2) if bias2 is not None else None), bias
        elif mode == "f32s" and (T * B) % 4 == 0 and I % 4 == 0 and h % 4 == 0:
            ctx.Ws = split_bf16x3(W_ih, 1, True)                              # [8h, 3I] (hi, lo, hi): kept for the backward's dX
            Gx, kbias = torch.mm(split_bf16x3(x.view(T * B, I), 1, False), ctx.Ws.t(), out_dtype=torch.float32), bias
        else:
            Gx, kbias = _mm(x.view(T * B, I), W_ih.t(), mode), bias           # bias added inside the recurrence kernel
        out = torch.empty((B, T, 2 * h) if bm else (T, B, 2 * h), device=x.device, dtype=torch.float32)
        R = torch.empty(T, 2, B, h, 4, device=x.device, dtype=torch.float32)
        Cs = torch.empty(T, 2, B, h, device=x.device, dtype=torch.float32)
        sync, nws = _lstm_fwd_ws(B, T, h, x.device)                        # sync words + the persistent kernel's exchange ring
        # outside the strict-fp32 mode the recurrence's W_hh products are split-precision bf16 MFMAs as well (TSG_F32S)
        ctx.rec_dtype = TSG_F32 if mode is None else TSG_F32S
        check_lstm_errors()
        _call("tsg_lstm_fwd_ws", x, ptr(Gx), ptr(kbias) if kbias is not None else None, ptr(W_hh), ptr(out), ptr(R), ptr(Cs),
              ptr(sync), nws, B, T, h, ctx.rec_dtype, int(bm))
        ctx.lstm_sync = sync
        ctx.bm = bool(bm)
        ctx.save_for_backward(x, W_ih, W_hh, out, R, Cs)
        ctx.mark_non_differentiable(Cs)
        ctx.set_materialize_grads(False)            # autograd would otherwise zero-fill a Cs-sized gradient every backward
        return out, Cs

    @staticmethod
    @_bwd
    def backward(ctx, dOut, _dCs):
        x, W_ih, W_hh, out, R, Cs = ctx.saved_tensors
        bm = ctx.bm
        (B, T, I) = x.shape if bm else (x.shape[1], x.shape[0], x.shape[2])
        h = W_hh.shape[2]
        TB = T * B
        dOut = _f32c(dOut) if dOut is not None else torch.zeros_like(out)
        WhhT = transposed(W_hh)
        dG = torch.empty((B, T, 2, 4 * h) if bm else (T, B, 2, 4 * h), device=x.device, dtype=torch.float32)
        dC = torch.empty(2, B, h, device=x.device, dtype=torch.float32)
        nb = int(load().tsg_lstm_bwd_ws_bytes(B, T, h))                      # ring workspace of the persistent backward (0: none)
        ws = torch.empty(nb // 4 + 4, device=x.device, dtype=torch.float32) if nb > 0 else None
        fused_db = ws is not None and bool(load().tsg_lstm_bwd_ws_persistent(B, T, h, nb))   # persistent path also sums dG -> dbias
        dbias = torch.empty(8 * h, device=x.device, dtype=torch.float32) if fused_db else None
        check_lstm_errors()
        _gate_before()
        _call("tsg_lstm_bwd_ws_layout", x, ptr(WhhT), ptr(R), ptr(Cs), ptr(dOut), None, ptr(dG), ptr(dC),
              ptr(ws) if ws is not None else None, nb, ptr(dbias) if fused_db else None, B, T, h, ctx.rec_dtype, int(bm))
        _gate_after()
        dGf = dG.view(TB, 8 * h)
        mode = ctx.mode
        fast = mode == "f32s" and T > 1 and TB % 16 == 0 and h % 4 == 0 and I % 4 == 0
        dx = _mm(dGf, W_ih, mode).view(x.shape) if (ctx.needs_input_grad[0] and not fast) else None
        if dbias is None:
            dbias = dGf.sum(0)
        # dW_hh[d] = sum_t dG_t[d]^T h_{t-1}[d] (h_{t+1} for the reverse direction): in row terms the partner of row r is
        # row r -+ B of `out` (time-major) or row r -+ 1 of the same sequence (batch-major), zero at the sequence ends
        shift, period = (1, T) if bm else (B, 0)
        x2, o2 = x.view(TB, I), out.view(TB, 2 * h)
        dx_gemm = gemm_f32s_ok(TB, I, 8 * h)
        dx_split = not dx_gemm and _LSTM_SPLITK and wgrad_f32s_ok(8 * h, TB, I)      # few rows, long contraction (the sentence encoder: 1280 rows, K = 4096)
        if ctx.own and T > 1 and wgrad_f32s_ok(TB, 4 * h, I, h) and (dx_gemm or dx_split or not ctx.needs_input_grad[0]):
            # own kernels, no operand planes: dX = dG W_ih on tsg_gemm_f32s (one transposed copy of the weight), and ONE weight-gradient
            # launch for dG[d]^T [x | h_{t-+1}[d]] of both directions, the shifted h rows read straight from `out`, written as the two
            # parameter-shaped tensors (tsg_wgrad_f32s_out2)
            if ctx.needs_input_grad[0]:
                if dx_gemm:
                    dx = gemm_f32s(dGf, transposed(W_ih)).view(x.shape)
                else:
                    # [1280 x 4096] . [4096 x 1024]: 20 output tiles cannot fill the chip, and the GEMM kernel has no split over K.  The
                    # weight-gradient kernel does (row ranges + a reduce pass) and takes both operands contraction-major: dX = (dG^T)^T W_ih
                    # with one transposed copy of dG (20 MB).  (Until round 5: bf16 planes + an 8-chunk library bmm + a sum: 71 + 12 + 36 us.)
                    dx = wgrad_f32s(transposed(dGf), W_ih)[0].view(x.shape)
            dW_ih, dW_hh = wgrad_f32s_out2(dGf, x2, o2, N=4 * h, K1=h, a_group_stride=4 * h, b1_group_stride=h, shift=shift, period=period)
            return dx, dW_ih.view(8 * h, I), dbias, dW_hh, None, None, (dbias if ctx.two_biases else None)
        if T == 1:
            dW_ih = _mm(dGf.t(), x2, mode)
            dW_hh = torch.zeros_like(W_hh)
        elif fast:
            # ONE batched GEMM over the two directions: D[d] = dG[d]^T [x | h_{t-+1}[d]] holds dW_ih[d] and dW_hh[d].  (Three
            # separate GEMMs had 199 / 80 / 80 output tiles and ran at 0.8 / 0.33 / 0.33 PFLOP/s, 1.11 ms per video layer;
            # one [8h x (I+2h)] GEMM that also computes the two cross-direction blocks: 0.59 ms; the 2-batch bmm without
            # them: 0.43 ms.)  Both operands are written K-contiguous by the transposing split, the layout the library's
            # bf16 GEMM runs this shape fastest in; the shifted h rows come straight from `out` (no shifted copies).
            At = torch.empty(8 * h, 3 * TB, device=x.device, dtype=torch.bfloat16)
            Bt = torch.empty(2 * (I + h), 3 * TB, device=x.device, dtype=torch.bfloat16)     # [2][I + h][3TB]
            split_bf16x3_t(dGf, 0, 8 * h, 0, False, At)
            if ctx.needs_input_grad[0]:
                # dX = dG W_ih from the SAME planes: At viewed as [(gate column, plane), T*B] is the transposed left operand
                # with the contraction ordered (column, plane); W_ih's (hi, lo, hi) planes in that order, K-contiguous, are a
                # 24 MB permute of its split.  No second pass over dG (110 us per layer).
                Ws = ctx.Ws if ctx.Ws is not None else split_bf16x3(W_ih, 1, True)
                Wt = Ws.view(8 * h, 3, I).permute(2, 0, 1).reshape(I, 24 * h)
                if TB <= 4096 and h % 64 == 0 and 24 * h >= 6144 and _LSTM_SPLITK:
                    # few rows, long contraction (the sentence encoder: 1280 x 12288 x I): the library runs it on 50 tiles at 0.3 PFLOP/s
                    # (98 + 103 us per step).  The same product as 8 K-chunks in one batched GEMM + a sum of the 8 partial results.
                    c, kc = 8, 3 * h
                    dx = torch.bmm(At.view(c, kc, TB).transpose(1, 2), Wt.view(I, c, kc).permute(1, 2, 0), out_dtype=torch.float32).sum(0).view(x.shape)
                else:
                    dx = torch.mm(At.view(24 * h, TB).t(), Wt.t(), out_dtype=torch.float32).view(x.shape)
            split_bf16x3_t(x2, 0, I, 0, True, Bt, 0, dup_row0=I + h)         # x planes, in both directions' batches
            split_bf16x3_t(o2, 0, h, shift, True, Bt, I, period)             # h_{t-1}, forward direction
            split_bf16x3_t(o2, h, h, -shift, True, Bt, 2 * I + h, period)    # h_{t+1}, reverse direction
            D = torch.bmm(At.view(2, 4 * h, 3 * TB), Bt.view(2, I + h, 3 * TB).transpose(1, 2), out_dtype=torch.float32)
            dW_ih = D[:, :, :I].contiguous().view(8 * h, I)                 # contiguous blocks: the per-parameter slices autograd
            dW_hh = D[:, :, I:].contiguous()                                # cuts from them are taken over as they are (no clones)
        else:
            # the step whose partner is the zero state drops out, so both operands are plain strided VIEWS
            dW_ih = _mm(dGf.t(), x2, mode)
            if bm:
                gf, hf = dG[:, 1:, 0].reshape(B * (T - 1), 4 * h), out[:, :-1, :h].reshape(B * (T - 1), h)
                gr, hr = dG[:, :-1, 1].reshape(B * (T - 1), 4 * h), out[:, 1:, h:].reshape(B * (T - 1), h)
            else:
                gf, hf = dG[1:, :, 0].reshape((T - 1) * B, 4 * h), out[:-1, :, :h].reshape((T - 1) * B, h)
                gr, hr = dG[:-1, :, 1].reshape((T - 1) * B, 4 * h), out[1:, :, h:].reshape((T - 1) * B, h)
            dW_hh = torch.stack([_mm(gf.t(), hf, mode), _mm(gr.t(), hr, mode)])
        return dx, dW_ih, dbias, dW_hh, None, None, (dbias if ctx.two_biases else None)


def lstm_bf16_ok(T: int, h: int) -> bool:
    """Shapes the persistent kernels take with dtype TSG_BF16 (include/tsg_hip.h)."""
    return T > 1 and h in (128, 256, 384, 512)


class _BiLSTMLayerBf16(torch.autograd.Function):
    """``_BiLSTMLayer`` in the bf16 storage mode: x, Gx = x W_ih^T, out, the saved gates R and the gradients dOut / dG / dX are
    bf16 tensors; the recurrence kernels run with dtype TSG_BF16 (one bf16 MFMA per k block, W_hh rounded once in the kernel's
    prologue, fp32 cell state, fp32 Cs); W_ih / W_hh / bias gradients leave their GEMMs / the kernel in fp32.  Batch-major."""

    @staticmethod
    def forward(ctx, x, W_ih, bias, W_hh, W_ih_bf16=None):
        require_device(x, W_ih, bias, W_hh)
        x, bias, W_hh = _bfc(x), _f32p(bias), _f32p(W_hh)
        B, T, I = x.shape
        h = W_hh.shape[2]
        if W_ih.shape != (8 * h, I) or W_hh.shape != (2, 4 * h, h) or bias.numel() != 8 * h:
            raise ValueError(f"bilstm: shape mismatch x{tuple(x.shape)} W_ih{tuple(W_ih.shape)} W_hh{tuple(W_hh.shape)}")
        Wb = W_ih_bf16 if (W_ih_bf16 is not None and W_ih_bf16.shape == W_ih.shape and W_ih_bf16.dtype == _BF) else weight_bf16(W_ih)     # [8h, I]
        Gx = _mm_bf16_nt(x.view(B * T, I), Wb)                            # bf16 [B*T, 8h]; the bias is added inside the kernel
        out = torch.empty(B, T, 2 * h, device=x.device, dtype=_BF)
        R = torch.empty(T, 2, B, h, 4, device=x.device, dtype=_BF)
        Cs = torch.empty(T, 2, B, h, device=x.device, dtype=torch.float32)
        sync, nws = _lstm_fwd_ws(B, T, h, x.device)
        check_lstm_errors()
        _call("tsg_lstm_fwd_ws", x, ptr(Gx), ptr(bias), ptr(W_hh), ptr(out), ptr(R), ptr(Cs), ptr(sync), nws, B, T, h, TSG_BF16, 1)
        ctx.lstm_sync = sync
        ctx.save_for_backward(x, Wb, W_hh, out, R, Cs)
        ctx.mark_non_differentiable(Cs)
        ctx.set_materialize_grads(False)
        return out, Cs

    @staticmethod
    def backward(ctx, dOut, _dCs):
        x, Wb, W_hh, out, R, Cs = ctx.saved_tensors
        B, T, I = x.shape
        h = W_hh.shape[2]
        TB = T * B
        dOut = _bfc(dOut) if dOut is not None else torch.zeros_like(out)
        WhhT = transposed(W_hh)
        dG = torch.empty(B, T, 2, 4 * h, device=x.device, dtype=_BF)
        dC = torch.empty(2, B, h, device=x.device, dtype=torch.float32)
        nb = int(load().tsg_lstm_bwd_ws_bytes(B, T, h))
        ws = torch.empty(nb // 4 + 4, device=x.device, dtype=torch.float32)
        dbias = torch.empty(8 * h, device=x.device, dtype=torch.float32)
        check_lstm_errors()
        _gate_before()
        _call("tsg_lstm_bwd_ws_layout", x, ptr(WhhT), ptr(R), ptr(Cs), ptr(dOut), None, ptr(dG), ptr(dC), ptr(ws), nb, ptr(dbias),
              B, T, h, TSG_BF16, 1)
        _gate_after()
        dGf = dG.view(TB, 8 * h)
        dx = _mm_bf16_nn(dGf, Wb).view(x.shape) if ctx.needs_input_grad[0] else None
        if _WGRAD_KERNEL and wgrad_f32s_ok(TB, 4 * h, I, h):
            # ONE launch per layer: D[d] = dG[d]^T [x | h_{t-+1}[d]] for both directions holds dW_ih[d] and dW_hh[d]; the shifted
            # h rows are read straight from `out` (row r -+ 1 of the same sequence, zero at its ends) -- no shifted copy, no fp32
            dW_ih, dW_hh = wgrad_bf16_out2(dGf, x.view(TB, I), out.view(TB, 2 * h), N=4 * h, K1=h, a_group_stride=4 * h, b1_group_stride=h,
                                           shift=1, period=T)            # the two parameter-shaped tensors: no slicing copies
            return dx, dW_ih.view(8 * h, I), dbias, dW_hh, None
        dW_ih = torch.mm(dGf.t(), x.view(TB, I), out_dtype=torch.float32)           # [8h, I], both directions
        # dW_hh[d] = sum_t dG_t[d]^T h_{t-1}[d] (h_{t+1} for the reverse direction): the partner rows as ONE shifted bf16 copy of
        # `out` (zero at the sequence ends), then a GEMM per direction on strided column views (no cat, no fp32 operands)
        hp = torch.empty_like(out)
        hp[:, 1:, :h] = out[:, :-1, :h]; hp[:, 0, :h] = 0
        hp[:, :-1, h:] = out[:, 1:, h:]; hp[:, -1, h:] = 0
        g2, hp2 = dG.view(TB, 2, 4 * h), hp.view(TB, 2 * h)
        dW_hh = torch.stack([torch.mm(g2[:, 0].t(), hp2[:, :h], out_dtype=torch.float32),
                             torch.mm(g2[:, 1].t(), hp2[:, h:], out_dtype=torch.float32)])
        return dx, dW_ih, dbias, dW_hh, None


def bilstm_layer(x, W_ih, bias, W_hh, batch_major=False, bias2=None, W_ih_bf16=None):
    """x [T,B,I] (or [B,T,I] with batch_major) -> (out in the same layout, Cs [T,2,B,h] cell states, not differentiable).
    ``bias2``: a second bias vector (nn.LSTM keeps b_ih and b_hh apart); the layer uses bias + bias2.
    bf16 storage mode: bf16 in and out through the TSG_BF16 recurrence kernels where they exist (T > 1, h in 128..512 step 128,
    batch-major); other shapes run the fp32-storage path on an fp32 copy and return bf16."""
    if bf16_storage() and x.is_cuda:
        T, h = (x.shape[1] if batch_major else x.shape[0]), W_hh.shape[2]
        if bias2 is not None:
            bias = bias + bias2
        if batch_major and lstm_bf16_ok(T, h):
            return _BiLSTMLayerBf16.apply(x, W_ih, bias, W_hh, W_ih_bf16)       # (W_ih_bf16: the joined parameters' live bf16 shadow, or None)
        # the fp32-storage layer in the f32s arithmetic: the mode is an ARGUMENT (saved in ctx for the backward), not a flip of the global
        out, Cs = _BiLSTMLayer.apply(x.float(), W_ih, bias, W_hh, batch_major, "f32s")
        return out.to(_BF), Cs
    return _BiLSTMLayer.apply(x, W_ih, bias, W_hh, batch_major, _AUTO, bias2)