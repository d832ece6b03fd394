"""ctypes binding of libtsg_hip.so (include/tsg_hip.h).  No CPU fallback: every call needs the
built library and device tensors, and raises otherwise."""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_longlong, c_uint64, c_void_p

import torch  # noqa: F401  (must be imported first: libtsg_hip.so reuses torch's libamdhip64.so.7)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TSG_HIP_LIB", os.path.join(_HERE, "libtsg_hip.so"))   # override: developer builds
TSG_F32, TSG_BF16, TSG_F32S = 0, 1, 2

_lib = None

# name -> argtypes (restype is int unless listed in _RESTYPE); mirrors include/tsg_hip.h
_P, _I = c_void_p, c_int
_SIGNATURES = {
    "tsg_version": [],
    "tsg_last_error": [],
    "tsg_scdm_attn_fwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "tsg_scdm_attn_bwd": [_P] * 11 + [c_longlong] + [_I] * 6 + [_P],
    "tsg_scdm_bwd_ws_bytes": [_I] * 6,
    "tsg_scdm_bwd_mode": [_I],
    "tsg_scdm_bwd_fused_ok": [_I] * 5,
    "tsg_error_sink": [_P],
    "tsg_time_next_launch": [c_int], "tsg_timed_launch_us": [c_int, _P],
    "tsg_error_word": [_P],
    "tsg_scdm_gate_fwd": [_P] * 8 + [_I] * 6 + [_P],
    "tsg_scdm_gate_bwd": [_P] * 15 + [c_longlong] + [_I] * 6 + [_P],
    "tsg_wgrad_f32s_ws_bytes": [c_longlong] + [_I] * 4,
    "tsg_wgrad_f32s": [_P, c_longlong, c_longlong, _P, c_longlong, _I, _P, c_longlong, c_longlong, _I, c_longlong, c_longlong,
                       _P, c_longlong, c_longlong, _P, c_longlong, c_longlong, _I, _I, _P],
    "tsg_wgrad_f32s_out2": [_P, c_longlong, c_longlong, _P, c_longlong, _I, _P, c_longlong, c_longlong, _I, c_longlong, c_longlong,
                            _P, c_longlong, c_longlong, _P, c_longlong, c_longlong, _P, c_longlong, c_longlong, _I, _I, _P],
    "tsg_wgrad_bf16_out2": [_P, c_longlong, c_longlong, _P, c_longlong, _I, _P, c_longlong, c_longlong, _I, c_longlong, c_longlong,
                            _P, c_longlong, c_longlong, _P, c_longlong, c_longlong, _P, c_longlong, c_longlong, _I, _I, _P],
    "tsg_wgrad_bf16": [_P, c_longlong, c_longlong, _P, c_longlong, _I, _P, c_longlong, c_longlong, _I, c_longlong, c_longlong,
                       _P, c_longlong, c_longlong, _P, c_longlong, c_longlong, _I, _I, _P],
    "tsg_boundary_score_fwd": [_P] * 9 + [_I] * 4 + [_P],
    "tsg_boundary_score_bwd": [_P] * 17 + [_I] * 4 + [_P],
    "tsg_boundary_score_bwd_ws_bytes": [_I] * 3,
    "tsg_boundary_score_bwd_ws": [_P] * 17 + [c_longlong] + [_I] * 4 + [_P],
    "tsg_mha_fwd": [_P] * 7 + [_I] * 6 + [c_float, _I, c_float, c_uint64, c_uint64, _I, _P],
    "tsg_mha_fwd_rng": [_P] * 7 + [_I] * 6 + [c_float, _I, c_float, _P, _I, _P],
    "tsg_mha_bwd_rng": [_P] * 10 + [_I] * 6 + [c_float, _I, c_float, _P, _I, _P],
    "tsg_lstm_fwd": [_P] * 6 + [_I] * 4 + [_P],
    "tsg_lstm_error_sink": [_P],
    "tsg_lstm_set_l2_exchange": [_I],
    "tsg_lstm_set_persist": [_I],
    "tsg_lstm_set_ring": [_I],
    "tsg_lstm_set_wide": [_I],
    "tsg_wgrad_set_stream_k": [_I],
    "tsg_adam_step": [_I, _P, _P, _P, _P, _P] + [ctypes.c_double] * 6 + [_P, _P, _P],
    "tsg_grads_nonfinite": [_I, _P, _P, _P, _P],
    "tsg_adam_step_shadow": [_I, _P, _P, _P, _P, _P, _P] + [ctypes.c_double] * 6 + [_P, _P, _P],
    "tsg_gemm_bf16": [_P, c_longlong, _P, c_longlong, _P, _P, c_longlong, _I, _I, _I, _I, _P],
    "tsg_match_head_fwd": [_P] * 5 + [_I] * 5 + [_P],
    "tsg_match_head_bwd": [_P] * 8 + [_I] * 5 + [_P],
    "tsg_gmd_losses_fwd": [_P] * 13 + [_I, _I, c_float, c_float, c_float, _P],
    "tsg_gmd_losses_bwd": [_P] * 20 + [_I, _I, c_float, c_float, c_float, _P],
    "tsg_lstm_fwd_bias": [_P] * 7 + [_I] * 5 + [_P],
    "tsg_lstm_fwd_ws": [_P] * 7 + [c_longlong] + [_I] * 5 + [_P],
    "tsg_lstm_fwd_ws_bytes": [_I, _I, _I],
    "tsg_lstm_bwd": [_P] * 7 + [_I] * 4 + [_P],
    "tsg_lstm_bwd_ws": [_P] * 8 + [c_longlong, _P] + [_I] * 4 + [_P],
    "tsg_lstm_bwd_ws_layout": [_P] * 8 + [c_longlong, _P] + [_I] * 5 + [_P],
    "tsg_lstm_bwd_ws_persistent": [_I, _I, _I, c_longlong],
    "tsg_lstm_bwd_ws_bytes": [_I, _I, _I],
    "tsg_linear_fwd": [_P, _P, _P, _P, _I, _I, _I, _I, _P],
    "tsg_gemm_f32s": [_P] * 4 + [_I] * 3 + [_P],
    "tsg_gemm_f32s_ld": [_P, c_longlong, _P, c_longlong, _P, _P, c_longlong, _I, _I, _I, _P],
    "tsg_gemm_f32s_nn": [_P, c_longlong, _P, _P, _I, c_longlong, _P, _P, c_longlong, _I, _I, _I, _P],
    "tsg_gemm_f32s_nn_acc": [_P, c_longlong, _P, _P, _I, c_longlong, _P, c_longlong, _I, _I, _I, _P],
    "tsg_head_gemm_ws_bytes": [_I, _I, _I],
    "tsg_match_head_gemm": [_P, c_longlong, _P, c_longlong, _P, _P, _P, _P, _P, _P, c_longlong, _I, _I, _I, _I, _I, _P],
    "tsg_boundary_head_gemm": [_P, c_longlong, _P, _P, c_longlong] + [_P] * 10 + [c_longlong, _I, _I, _I, _I, _P],
    "tsg_boundary_softmax": [_P, _P, _I, _I, _P],
    "tsg_transpose_f32": [_P, c_longlong, _P, _I, _I, _I, _P],
    "tsg_dropout": [_P, _P, c_longlong, c_float, c_uint64, c_uint64, _P, _P, _I, _I, _P],
    "tsg_layer_norm_fwd": [_P] * 6 + [c_longlong, _I, c_float, _I, _P],
    "tsg_layer_norm_bwd_ws_bytes": [c_longlong, _I],
    "tsg_layer_norm_bwd": [_P] * 9 + [c_longlong, c_longlong, _I, _I, _P],
    "tsg_moment_pool_fwd": [_P] * 5 + [_I] * 4 + [_P],
    "tsg_moment_pool_bwd": [_P] * 5 + [_I] * 4 + [_P],
    "tsg_split_bf16x3": [_P, _P, c_longlong, c_longlong, c_longlong, c_longlong, _I, _P],
    "tsg_split_bf16x3_shift": [_P, c_longlong, c_longlong, c_longlong, _P, c_longlong, c_longlong, c_longlong, c_longlong, _I, _P],
    "tsg_split_bf16x3_t": [_P, c_longlong, c_longlong, c_longlong, _P, c_longlong, c_longlong, c_longlong, c_longlong, _I, c_longlong, _P],
    "tsg_pool_clips": [_P] * 6 + [_I] * 4 + [_P],
    "tsg_sequence_masks": [_P] * 6 + [_I] * 2 + [_P],
    "tsg_moment_translate": [_P] * 4 + [c_uint64, _P, _P] + [_I] * 4 + [_P],
    "tsg_span_pred": [_P] * 4 + [_I] * 3 + [_P],
    "tsg_mha_bwd": [_P] * 10 + [_I] * 6 + [c_float, _I, c_float, c_uint64, c_uint64, _I, _P],
}
_RESTYPE = {"tsg_last_error": c_char_p, "tsg_lstm_bwd_ws_bytes": c_longlong, "tsg_lstm_fwd_ws_bytes": c_longlong, "tsg_scdm_bwd_ws_bytes": c_longlong,
             "tsg_wgrad_f32s_ws_bytes": c_longlong, "tsg_boundary_score_bwd_ws_bytes": c_longlong, "tsg_head_gemm_ws_bytes": c_longlong, "tsg_layer_norm_bwd_ws_bytes": c_longlong}


class TsgLibraryError(RuntimeError):
    pass


def exported_symbols():
    return sorted(_SIGNATURES)


def load() -> ctypes.CDLL:
    """Load (once) and return the library; raises TsgLibraryError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise TsgLibraryError(
            f"{LIB_PATH} not found: build it with `python -m shufflingvideosfortsg_amd.build` "
            "(or __graft_entry__.build()).  There is no CPU fallback for the hot path.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in _SIGNATURES.items():
        fn = getattr(lib, name)            # AttributeError here = header / library mismatch
        fn.argtypes = argtypes
        fn.restype = _RESTYPE.get(name, c_int)
    if lib.tsg_version() != 7:
        raise TsgLibraryError(f"libtsg_hip.so version {lib.tsg_version()} != 7 expected by the Python host code")
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().tsg_last_error()
        raise RuntimeError(f"{what} failed (rc={rc}): {msg.decode() if msg else '?'}")


def ptr(t: torch.Tensor) -> int:
    return t.data_ptr()


def stream_of(t: torch.Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream


def require_device(*tensors: torch.Tensor) -> None:
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                "shufflingvideosfortsg_amd hot-path ops run only on an MI355X through libtsg_hip.so; "
                f"got a {t.device} tensor (there is no CPU fallback)")
