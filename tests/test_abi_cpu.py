"""The C-ABI library: builds, loads, exports exactly what include/tsg_hip.h declares, and rejects
bad arguments before touching a device (no GPU needed)."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from shufflingvideosfortsg_amd import _lib, build
    build.build()                         # hipcc cross-compiles gfx950 without a GPU
    return _lib.load()


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "tsg_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(tsg_[a-z0-9_]+)\s*\(", src)))


def test_exports_match_header(lib):
    from shufflingvideosfortsg_amd import _lib
    want = declared_symbols()
    assert want == _lib.exported_symbols(), "ctypes signature table and header disagree"
    for name in want:
        assert hasattr(lib, name), f"{name} declared in tsg_hip.h but not exported by libtsg_hip.so"
    assert lib.tsg_version() == 7


def test_argument_errors_without_gpu(lib):
    from shufflingvideosfortsg_amd._lib import TSG_F32
    buf = (ctypes.c_float * 64)()
    p = ctypes.addressof(buf)
    p = (p + 15) & ~15
    # NULL pointer
    assert lib.tsg_scdm_attn_fwd(None, p, p, p, p, p, 1, 1, 1, 4, 4, TSG_F32, None) == -1
    assert b"NULL" in lib.tsg_last_error()
    # bad shapes / dtype / alignment
    assert lib.tsg_scdm_attn_fwd(p, p, p, p, p, p, 1, 1, 33, 4, 4, TSG_F32, None) == -2      # N > 32
    assert lib.tsg_scdm_attn_fwd(p, p, p, p, p, p, 1, 1, 1, 6, 4, TSG_F32, None) == -3       # H % 4
    assert lib.tsg_scdm_attn_fwd(p, p, p, p, p, p, 1, 1, 1, 4, 4, 7, None) == -4             # dtype
    assert lib.tsg_scdm_attn_fwd(p + 4, p, p, p, p, p, 1, 1, 1, 4, 4, TSG_F32, None) == -3   # misaligned
    assert lib.tsg_mha_fwd(p, p, p, p, None, None, p, 1, 4, 4, 8, 8, 3, 1.0, 0, 0.0, 0, 0, TSG_F32, None) == -2  # 8 % 3
    assert lib.tsg_mha_fwd(p, p, p, p, None, None, p, 1, 4, 4, 8, 8, 2, 1.0, 0, 1.0, 0, 0, TSG_F32, None) == -2  # dropout probability outside [0,1)
    assert lib.tsg_boundary_score_fwd(p, p, p, p, p, None, None, p, p, 1, 4, 3, TSG_F32, None) == -2             # 2*Hm % 4
    assert lib.tsg_mha_bwd(p, p, p, p, p, p, p, p, p, p, 1, 4, 4, 8, 8, 2, 1.0, 0, 0.0, 0, 0, 7, None) == -4     # dtype (TSG_F32 / TSG_F32S only)
    # dtype TSG_BF16 (bf16 storage, ABI revision 3): accepted where the header says so, shapes it does not take are TSG_E_SHAPE (-2)
    from shufflingvideosfortsg_amd._lib import TSG_BF16
    assert lib.tsg_mha_fwd(p, p, p, p, p, None, p, 1, 4, 4, 64, 64, 2, 1.0, 0, 0.0, 0, 0, TSG_BF16, None) == -2      # A_sum asked for
    assert lib.tsg_mha_fwd(p, p, p, p, None, None, p, 1, 4, 4, 40, 40, 2, 1.0, 0, 0.0, 0, 0, TSG_BF16, None) == -2   # head width 20
    assert lib.tsg_mha_bwd(p, p, p, p, p, p, p, p, p, p, 1, 4, 4, 512, 512, 2, 1.0, 0, 0.0, 0, 0, TSG_BF16, None) == -2   # head width 256
    assert lib.tsg_lstm_fwd_bias(p, None, p, p, p, p, p, 4, 16, 100, TSG_BF16, 1, None) == -2                      # h = 100: no persistent kernel
    assert lib.tsg_lstm_fwd_bias(p, None, p, p, p, p, None, 4, 16, 128, TSG_BF16, 1, None) == -2                   # no sync workspace
    assert lib.tsg_lstm_bwd_ws_layout(p, p, p, p, None, p, p, None, 0, None, 4, 16, 128, TSG_BF16, 1, None) == -2  # no ring workspace
    assert lib.tsg_match_head_fwd(p, p, p, p, p, 1, 4, 8, 0, 7, None) == -4                                        # dtype (new argument)
    assert lib.tsg_boundary_score_fwd(p, p, p, p, p, None, None, p, p, 1, 4, 4, 7, None) == -4
    assert lib.tsg_error_word(None) == 0 and lib.tsg_error_sink(None) == 0                                         # un-registering is allowed
    assert lib.tsg_wgrad_bf16(p, 256, 0, p, 100, 128, None, 0, 0, 0, 0, 0, p, 128, 0, None, 0, 64, 256, 1, None) == -2    # ldb0 < K0
    # the weight-gradient GEMM: host-side plan and argument checks (no launch)
    assert lib.tsg_wgrad_f32s_ws_bytes(16384, 1024, 1024, 0, 1) == 8 * 4 * 1024 * 1024      # 32 tiles -> 8 row ranges of partial tiles
    assert lib.tsg_wgrad_f32s_ws_bytes(16384, 2048, 1024, 512, 2) == 4 * 4 * 2 * 2048 * 1536  # the LSTM shape: 192 tiles -> 4 ranges
    assert lib.tsg_wgrad_f32s_ws_bytes(32, 256, 128, 0, 1) == 0                               # one chunk: no partials
    assert lib.tsg_wgrad_f32s_ws_bytes(48, 256, 128, 0, 1) == -1                              # M % 32
    assert lib.tsg_wgrad_f32s_ws_bytes(64, 128, 128, 0, 1) == -1                              # N % 256
    assert lib.tsg_wgrad_f32s(p, 256, 0, p, 128, 128, None, 0, 0, 0, 0, 0, p, 128, 0, None, 0, 64, 256, 3, None) == -2   # groups
    assert lib.tsg_wgrad_f32s(p, 256, 0, p, 100, 128, None, 0, 0, 0, 0, 0, p, 128, 0, None, 0, 64, 256, 1, None) == -2   # ldb0 < K0
    assert lib.tsg_wgrad_f32s(None, 256, 0, p, 128, 128, None, 0, 0, 0, 0, 0, p, 128, 0, None, 0, 64, 256, 1, None) == -1  # NULL


def test_no_cpu_fallback():
    """Hot-path ops and modules refuse CPU tensors instead of silently computing elsewhere."""
    from shufflingvideosfortsg_amd import functional as F
    from shufflingvideosfortsg_amd.model.networks.attention import SCDM_Attention, MultiHead
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        F.scdm_attn(torch.randn(1, 4, 8), torch.randn(1, 3, 8), torch.randn(8), torch.randn(1, 3, 8))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        SCDM_Attention(8, 8)(torch.randn(1, 4, 8), torch.randn(1, 3, 8))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        MultiHead(8, 8, 2, 0.0)(torch.randn(1, 4, 8), torch.randn(1, 4, 8), torch.randn(1, 4, 8))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        F.boundary_score(torch.randn(1, 4, 8), torch.randn(1, 8), torch.randn(8), torch.randn(8), torch.randn(2))


def test_product_does_not_import_oracle():
    """oracle/ is test infrastructure: nothing under the package may reference it."""
    pkg = os.path.join(ROOT, "shufflingvideosfortsg_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(d, f), errors="ignore").read()
                assert "oracle" not in txt.replace("test oracle", ""), f"{f} mentions the oracle"


def test_timing_hook_argument_errors(lib):
    """tsg_time_next_launch / tsg_timed_launch_us (ABI 7): slot range and never-used slots are argument errors; -1 disarms."""
    us = ctypes.c_float(-1.0)
    assert lib.tsg_time_next_launch(4096) == -2 and b"slot" in lib.tsg_last_error()
    assert lib.tsg_time_next_launch(-1) == 0
    assert lib.tsg_timed_launch_us(7, ctypes.byref(us)) == -2          # no launch used slot 7
    assert lib.tsg_timed_launch_us(7, None) == -1
    assert lib.tsg_time_next_launch(7) == 0 and lib.tsg_time_next_launch(-1) == 0
    assert lib.tsg_timed_launch_us(7, ctypes.byref(us)) == -2          # armed and disarmed without a launch


def test_grads_nonfinite_argument_errors(lib):
    buf = (ctypes.c_float * 64)()
    p = (ctypes.addressof(buf) + 15) & ~15
    one = (ctypes.c_void_p * 1)(p); cnt = (ctypes.c_longlong * 1)(8)
    assert lib.tsg_grads_nonfinite(0, one, cnt, p, None) == -1
    assert lib.tsg_grads_nonfinite(1, one, cnt, None, None) == -1
    assert lib.tsg_grads_nonfinite(1, (ctypes.c_void_p * 1)(None), cnt, p, None) == -1
    assert lib.tsg_grads_nonfinite(1, one, (ctypes.c_longlong * 1)(0), p, None) == -2
    assert lib.tsg_grads_nonfinite(1, (ctypes.c_void_p * 1)(p + 2), cnt, p, None) == -3
