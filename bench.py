#!/usr/bin/env python3
"""Benchmark of the cross-modal matching hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

N>1: one rank per GPU over RCCL.  Either the caller launches the ranks (``python -m torch.distributed.run
--nproc-per-node N ... bench.py --gpus N``: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* come from the environment) or,
when WORLD_SIZE is not set, this process starts them itself as CHILD processes -- before it has imported torch or touched
a GPU -- relays rank 0's one JSON line and exits with the launcher's code.  A world size different from --gpus is an error.

One "step" = one full training step of the grounding model over one synthetic batch of B=64
clip-query pairs per GPU at [T_clip=128, T_word=20, d=1024] (BASELINE.json metric shape), fp32:
forward -> losses -> backward -> gradient all-reduce (RCCL, N>1) -> Adam.  Default model = GMD, the
shuffling framework's train step (original + shuffled video, BASELINE config 2); ``--model qave`` runs
the QAVE baseline step.  Inputs are generated on the host and made resident in HBM before the timed
region.  Weak scaling: per-GPU batch fixed, `value` = pairs of ALL ranks / max-over-ranks time.

The one JSON line also carries
  roofline     : the cross-attention kernel as the step launches it (K1 forward with the gate epilogue,
                 tsg_scdm_gate_fwd; the plain tsg_scdm_attn_fwd for a model without the gate): algorithmic
                 bytes per launch / its mean duration measured with events on the launch stream INSIDE the
                 timed steps, against the 8 TB/s HBM3E peak;
  kernels      : the same for the other hot-path kernels (informational);
  cpu_baseline : the CPU oracle (oracle/tsg_oracle.py, a port of the reference's op graph) running
                 the same step on a bounded sample of the same workload on this host's cores.
"""
from __future__ import annotations

import argparse
import json
import re
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--model", default="gmd", choices=["gmd", "qave"])
    ap.add_argument("--B", type=int, default=64, help="clip-query pairs per GPU (--scaling weak, default) or in the GLOBAL batch (--scaling strong)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: B pairs per GPU, the job grows with N (default; `value` = N*B*steps / time).  strong: B is the GLOBAL batch, "
                         "every rank takes its contiguous B/N shard (dp.shard_batch: what the reference's DataParallel does with one batch, "
                         "train.py:343) -- BASELINE configs 3 / 4 are global batches of 64 / 128 over 4 / 8 GPUs = 16 pairs per GPU")
    ap.add_argument("--graph", default="auto", choices=["auto", "on", "off"],
                    help="replay the train step from two HIP graphs (engine.GraphedTrainStep) in THIS process and report it next to the "
                         "eager step; auto = on (off with --no-alt at N = 1).  The faster of the two is `value` and `value_mode` says which: "
                         "it is the same full step either way (with eight ranks on one host the replay's 0.15 instead of 12 ms of host time "
                         "per step is what keeps the ranks from queueing behind their Python threads)")
    ap.add_argument("--T", type=int, default=128)
    ap.add_argument("--N", type=int, default=20)
    ap.add_argument("--d", type=int, default=1024)
    ap.add_argument("--cpu-sample", type=int, default=32, help="pairs in the CPU-baseline sample (0 = skip); bounded to ~30 s")
    ap.add_argument("--no-micro", action="store_true", help="skip the stand-alone K1 / K2 launches")
    ap.add_argument("--dtype", default="f32s", choices=["f32", "bf16", "f32s", "bf16g"],
                    help="f32s (default, the fp32-parity headline): fp32 storage, matrix products as split-precision bf16 MFMA products, "
                         "hi*hi+hi*lo+lo*hi with fp32 accumulate -- error at the fp32 GEMM's level, the parity suite passes at the fp32 "
                         "tolerances; f32: strict fp32 (rocBLAS fp32 GEMMs, fp32 MFMA recurrence); bf16: the bf16 STORAGE path that BASELINE "
                         "configs 2 / 4 name -- activations and their gradients live in HBM as bf16 through every hand-written kernel "
                         "(dtype TSG_BF16), plain bf16 GEMMs, fp32 arithmetic / master weights / optimizer (~1e-2 relative to the fp32 "
                         "reference, tests/test_bf16_storage_gpu.py); bf16g: the older operands-only mode (fp32 storage, library GEMM "
                         "operands cast to bf16)")
    ap.add_argument("--predictor", default="mlp", choices=["mlp", "self_attn"],
                    help="boundary head: mlp (reference default, K3) or self_attn (temporal self-attention, K2 in the step)")
    ap.add_argument("--fwd-only", action="store_true",
                    help="a step is the forward pass + losses under no_grad (BASELINE config 1 is forward-only); `value` is then "
                         "forward pairs/s.  Without the flag the forward-only rate is a side measurement (`fwd_only`)")
    ap.add_argument("--graph-only", action="store_true", help=argparse.SUPPRESS)     # child mode of the graph_replay measurement
    ap.add_argument("--time-all", action="store_true", help="event-time every C-ABI launch (perturbs the step time)")
    ap.add_argument("--no-alt", action="store_true", help="skip the side measurements (other GEMM-operand modes, forward-only)")
    return ap.parse_args(argv)


def spawn_ranks(a):
    """--gpus N > 1 without a launcher environment: start N ranks as children of THIS process through
    torch.distributed.run (rendezvous on 127.0.0.1, a free port), relay the one JSON line rank 0 prints and return the
    launcher's exit code.  Nothing here imports torch or touches a GPU: a process that has initialised the GPU must never
    be replaced or re-executed, and this parent never initialises it."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL needs it on this driver
    print(f"[bench] starting {a.gpus} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    if p.returncode != 0 or len(lines) != 1:
        sys.stderr.write(p.stdout)
        print(f"[bench] launcher exit code {p.returncode}, {len(lines)} result lines", file=sys.stderr, flush=True)
        return p.returncode or 1
    got = json.loads(lines[0])
    if got.get("n_gpus") != a.gpus or got.get("rccl_ranks") != a.gpus:
        print(f"[bench] asked for {a.gpus} GPUs, result says n_gpus={got.get('n_gpus')} rccl_ranks={got.get('rccl_ranks')}",
              file=sys.stderr, flush=True)
        return 1
    print(lines[0], flush=True)
    return 0


if __name__ == "__main__" and "WORLD_SIZE" not in os.environ:
    _a = parse_args()
    if _a.gpus > 1:
        sys.exit(spawn_ranks(_a))

import shufflingvideosfortsg_amd          # noqa: E402,F401  (before torch touches the GPU: _runtime_env sets what the ROCm runtime must see at its start)
import torch                              # noqa: E402
import torch.distributed as dist          # noqa: E402

from shufflingvideosfortsg_amd import data, engine, functional  # noqa: E402
from shufflingvideosfortsg_amd.dp import FlatGradAllReduce      # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md)
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak (MI355X_MICROARCH.md; no sparsity)
MFMA_F32_PEAK_TFLOPS = 157.3    # fp32-input MFMA peak = the fp32 vector rate (MI355X_MICROARCH.md)
_T0 = time.time()


# stdout carries exactly ONE line, the JSON result: fd 1 is pointed at stderr for everything else that may print there
# (RCCL prints a five-line version banner on stdout when a communicator is created), the result goes to the saved fd.
_RESULT_FD = os.dup(1)
os.dup2(2, 1)


def log(msg):
    """progress on stderr (stdout carries exactly one JSON line)"""
    if int(os.environ.get("RANK", 0)) == 0:
        print(f"[bench {time.time() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


ALG_FORMULA = {     # per pair; e = bytes per stored activation element (4, or 2 with dtype TSG_BF16); P and the [B,T] vectors are fp32
    "scdm_fwd": "(2T+2N)*d*e + T*N*4   [SURVEY 8d: read a, s, sent; write C, P]",
    "scdm_bwd": "(3T+4N)*d*e + T*N*4   [SURVEY 8d: read a, s, sent, dC, P; write da, ds, dsent]",
    "scdm_gate_fwd": "(3T+2N)*d*e + T*N*4   [gate-fused K1g, NOT SURVEY 8d's plain-K1 figure: read a, r (2T), s, VW (2N); write out (T), P]",
    "scdm_gate_bwd": "(5T+4N)*d*e + T*N*4   [gate-fused: read a, r, dout (3T), s, VW (2N), P; write da, dr (2T), ds, dVW (2N)]",
    "boundary_fwd": "T*2Hm*e + 2T*4   [read y; write p_start, p_end]",
    "boundary_bwd": "2*T*2Hm*e + 4T*4   [read y; write dy]",
}


def alg_bytes(kind, B, T, N, d, Hm=256, e=4):
    """Algorithmic HBM bytes per launch (SURVEY.md 8d / DESIGN.md; formulas in ALG_FORMULA)."""
    if kind == "scdm_fwd":
        return B * ((2 * T + 2 * N) * d * e + T * N * 4)
    if kind == "scdm_bwd":
        return B * ((3 * T + 4 * N) * d * e + T * N * 4)
    if kind == "scdm_gate_fwd":          # reads a, r (2T), s, VW (2N); writes out (T), P
        return B * ((3 * T + 2 * N) * d * e + T * N * 4)
    if kind == "scdm_gate_bwd":          # reads a, r, dout (3T), s, VW (2N), P; writes da, dr (2T), ds, dVW (2N)
        return B * ((5 * T + 4 * N) * d * e + T * N * 4)
    if kind == "boundary_fwd":
        return B * (T * 2 * Hm * e + 2 * T * 4)
    if kind == "boundary_bwd":
        return B * (2 * T * 2 * Hm * e + 4 * T * 4)
    raise KeyError(kind)


def micro_kernels(B, T, N, d, heads=8, iters=30):
    """Stand-alone launches of the hot-path kernels the train step does not exercise (K2 = the
    multi-head attention of MultiHead / Self_Attention_predictor) at the bench shape: the C entry
    points are called back to back on preallocated buffers (no Python allocation between launches, so
    the GPU, not the host, sets the pace) and timed with one event pair on the launch stream.
    Outside the timed region; informational."""
    import math
    from shufflingvideosfortsg_amd import _lib
    from shufflingvideosfortsg_amd._lib import TSG_F32, ptr
    lib, dev = _lib.load(), "cuda"
    stream = torch.cuda.current_stream()
    st = stream.cuda_stream
    out = {}

    def timed_us(fns):
        for i in range(max(3, len(fns))):
            fns[i % len(fns)]()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for i in range(iters):
            fns[i % len(fns)]()
        e1.record(stream); e1.synchronize()
        return e0.elapsed_time(e1) * 1e3 / iters

    def run(name, fn, nbytes):
        """fn: one callable, or a LIST of callables on disjoint operand sets that are launched in rotation (round-4 review: thirty launches on
        the same 78 / 134 MB sit inside the 256 MB Infinity Cache, so their "HBM fraction" was cache-assisted).  With a list, `mean_us` /
        `frac` are the ROTATING figures (the sets together exceed the cache: every launch streams from HBM) and `cached_*` the old
        same-buffers figures, printed beside them."""
        fns = fn if isinstance(fn, (list, tuple)) else [fn]
        us = timed_us(fns)
        out[name] = {"mean_us": round(us, 2), "launches": iters, "alg_bytes": nbytes,
                     "achieved_GBs": round(nbytes / us / 1e3, 1), "frac": round(nbytes / us / 1e3 / HBM_PEAK_GBS, 4)}
        if len(fns) > 1:
            cus = timed_us(fns[:1])
            out[name].update(operand_sets=len(fns), operand_bytes_in_rotation=len(fns) * nbytes, cached_mean_us=round(cus, 2),
                             cached_frac=round(nbytes / cus / 1e3 / HBM_PEAK_GBS, 4),
                             note="frac = launches rotating over disjoint operand sets (> 256 MB in all: HBM); cached_frac = the same "
                                  "launch repeated on one set (inside the Infinity Cache)")

    def nsets(nbytes):                                       # operand sets so that the rotation exceeds the 256 MB Infinity Cache (>= 4)
        return max(4, -(-(320 << 20) // max(nbytes, 1)))
    e, sc = 4, math.sqrt(d)
    # K1 without the gate epilogue (the SCDM_Attention module on its own)
    w = torch.randn(d, device=dev) / sc; dw = torch.empty_like(w)
    nb = int(lib.tsg_scdm_bwd_ws_bytes(B, T, N, d, d, 0)); ws = torch.empty(nb // 4 + 4, device=dev)
    keep = []                                                # the operand sets stay alive while their closures are in use

    def k1_set():
        A = torch.randn(B, T, d, device=dev); S = torch.randn(B, N, d, device=dev)
        V = torch.randn(B, N, d, device=dev); C = torch.empty(B, T, d, device=dev); P = torch.empty(B, T, N, device=dev)
        dC = torch.randn(B, T, d, device=dev); da, ds, dV = torch.empty_like(A), torch.empty_like(S), torch.empty_like(V)
        keep.append((A, S, V, C, P, dC, da, ds, dV))
        return (lambda: lib.tsg_scdm_attn_fwd(ptr(A), ptr(S), ptr(w), ptr(V), ptr(C), ptr(P), B, T, N, d, d, TSG_F32, st),
                lambda: lib.tsg_scdm_attn_bwd(ptr(A), ptr(S), ptr(w), ptr(V), ptr(P), ptr(dC), ptr(da), ptr(ds), ptr(dw), ptr(dV), ptr(ws), nb,
                                              B, T, N, d, d, TSG_F32, st))
    k1 = [k1_set() for _ in range(nsets(alg_bytes("scdm_fwd", B, T, N, d)))]
    run(f"tsg_scdm_attn_fwd[alone: {B},{T},{N},{d}]", [f for f, _ in k1], alg_bytes("scdm_fwd", B, T, N, d))
    run(f"tsg_scdm_attn_bwd[alone: {B},{T},{N},{d}]", [b for _, b in k1], alg_bytes("scdm_bwd", B, T, N, d))
    del k1; keep.clear()
    for tag, Tk in (("cross", N), ("self", T)):
        def k2_set(dt):
            Q = torch.randn(B, T, d, device=dev); K = torch.randn(B, Tk, d, device=dev); V = torch.randn(B, Tk, d, device=dev)
            O = torch.empty(B, T, d, device=dev); lse = torch.empty(B, heads, T, device=dev); g = torch.randn(B, T, d, device=dev)
            dQ, dK, dV, dlt = torch.empty_like(Q), torch.empty_like(K), torch.empty_like(V), torch.empty_like(lse)
            keep.append((Q, K, V, O, lse, g, dQ, dK, dV, dlt))
            return (lambda: lib.tsg_mha_fwd(ptr(Q), ptr(K), ptr(V), ptr(O), None, None, ptr(lse), B, T, Tk, d, d, heads, sc, 0, 0.0, 0, 0, dt, st),
                    lambda: lib.tsg_mha_bwd(ptr(Q), ptr(K), ptr(V), ptr(O), ptr(g), ptr(lse), ptr(dQ), ptr(dK), ptr(dV), ptr(dlt), B, T, Tk, d, d,
                                            heads, sc, 0, 0.0, 0, 0, dt, st))
        fb, bb = B * (2 * T + 2 * Tk) * d * e, B * (4 * T + 4 * Tk) * d * e          # forward: read Q,K,V write O; backward: read Q,K,V,O,dO write dQ,dK,dV
        k2 = [k2_set(TSG_F32) for _ in range(nsets(fb))]
        for f, _ in k2:                                      # the backward reads the forward's O / lse of ITS set
            f()
        # matrix work: forward = QK^T + PV = 4 B Tq Tk d flops, backward = five such products = 10 B Tq Tk d.  The exact kernels
        # issue it on the fp32 MFMA (157 TFLOP/s: at ~40 flop/B they sit above that pipe's ridge, so their bound is the fp32 MFMA peak,
        # reported next to the HBM fraction); the split-precision kernels issue 3 bf16 products per fp32 product
        ffl, bfl = 4.0 * B * T * Tk * d, 10.0 * B * T * Tk * d

        def mfma(name, flops, peak, label):
            out[name].update({"mfma_flops": flops, "achieved_TFLOPs": round(flops / out[name]["mean_us"] / 1e6, 1),
                              "mfma_frac": round(flops / out[name]["mean_us"] / 1e6 / peak, 4), "mfma_peak": label})
        n = f"tsg_mha_fwd[{tag}: {B},{T},{Tk},{d},h{heads}]"
        run(n, [f for f, _ in k2], fb)
        mfma(n, ffl, MFMA_F32_PEAK_TFLOPS, "fp32 MFMA 157.3 TFLOP/s")
        n = f"tsg_mha_bwd[{tag}: {B},{T},{Tk},{d},h{heads}]"
        run(n, [b for _, b in k2], bb)
        mfma(n, bfl, MFMA_F32_PEAK_TFLOPS, "fp32 MFMA 157.3 TFLOP/s")
        del k2; keep.clear()
        # the split-precision kernels (dtype TSG_F32S: what the "f32s" mode launches)
        k2 = [k2_set(2) for _ in range(nsets(fb))]
        for f, _ in k2:
            f()
        n = f"tsg_mha_fwd[{tag}, f32s: {B},{T},{Tk},{d},h{heads}]"
        run(n, [f for f, _ in k2], fb)
        mfma(n, 3 * ffl, MFMA_BF16_PEAK_TFLOPS, "bf16 MFMA 2500 TFLOP/s, 3 products per fp32 product")
        n = f"tsg_mha_bwd[{tag}, f32s: {B},{T},{Tk},{d},h{heads}]"
        run(n, [b for _, b in k2], bb)
        mfma(n, 3 * bfl, MFMA_BF16_PEAK_TFLOPS, "bf16 MFMA 2500 TFLOP/s, 3 products per fp32 product")
        del k2; keep.clear()
    # the hand-written weight-gradient GEMM (csrc/wgrad_split.hip) at the step's three shapes: MFMA-bound, so its roofline is the
    # dense bf16 MFMA peak; flops = the bf16 matrix work it issues (3 products per fp32 product)
    for (M, Nn, Kk) in ((2 * B * T, d, d), (2 * B * N, d, d), (B * T, 512, d)):
        if M % 32 or Nn % 256 or Kk % 128:
            continue
        Aw = torch.randn(M, Nn, device=dev); Bw = torch.randn(M, Kk, device=dev); Cw = torch.empty(Nn, Kk, device=dev)
        nbw = int(lib.tsg_wgrad_f32s_ws_bytes(M, Nn, Kk, 0, 1)); wsw = torch.empty(max(nbw, 16), device=dev, dtype=torch.uint8)
        name = f"tsg_wgrad_f32s[{Nn}x{M} . {M}x{Kk}]"
        run(name, lambda: lib.tsg_wgrad_f32s(ptr(Aw), Nn, 0, ptr(Bw), Kk, Kk, None, 0, 0, 0, 0, 0, ptr(Cw), Kk, 0, ptr(wsw), nbw, M, Nn, 1, st),
            (M * (Nn + Kk) + Nn * Kk) * e)
        fl = 3 * 2.0 * M * Nn * Kk
        out[name].update(bound="mfma", mfma_flops=fl, achieved_TFLOPs=round(fl / out[name]["mean_us"] / 1e6, 1),
                         mfma_frac=round(fl / out[name]["mean_us"] / 1e6 / MFMA_BF16_PEAK_TFLOPS, 4))
    # the LSTM layer's weight gradient: both directions' dG[d]^T [x | h_(t-+1)[d]] in one launch, the shifted h rows read from the layer
    # output, two parameter-shaped outputs (tsg_wgrad_f32s_out2): 4 launches per step, the largest share of the weight-gradient time
    hh = d // 2
    if (2 * B * T) % 32 == 0 and (4 * hh) % 256 == 0 and d % 128 == 0 and hh % 128 == 0:
        M = 2 * B * T
        dGw = torch.randn(M, 8 * hh, device=dev); xw = torch.randn(M, d, device=dev); ow = torch.randn(M, 2 * hh, device=dev)
        C0 = torch.empty(2, 4 * hh, d, device=dev); C1 = torch.empty(2, 4 * hh, hh, device=dev)
        nbw = int(lib.tsg_wgrad_f32s_ws_bytes(M, 4 * hh, d, hh, 2)); wsw = torch.empty(max(nbw, 16), device=dev, dtype=torch.uint8)
        name = f"tsg_wgrad_f32s_out2[LSTM layer: 2 x {4 * hh}x{M} . {M}x({d}+{hh})]"
        run(name, lambda: lib.tsg_wgrad_f32s_out2(ptr(dGw), 8 * hh, 4 * hh, ptr(xw), d, d, ptr(ow), 2 * hh, hh, hh, 1, T, ptr(C0), d, 4 * hh * d,
                                                  ptr(C1), hh, 4 * hh * hh, ptr(wsw), nbw, M, 4 * hh, 2, st),
            (M * (8 * hh + d + 2 * hh) + 2 * 4 * hh * (d + hh)) * e)
        fl = 3 * 2.0 * M * 8 * hh * (d + hh)
        out[name].update(bound="mfma", mfma_flops=fl, achieved_TFLOPs=round(fl / out[name]["mean_us"] / 1e6, 1),
                         mfma_frac=round(fl / out[name]["mean_us"] / 1e6 / MFMA_BF16_PEAK_TFLOPS, 4))
    # the hand-written split-on-load projection GEMM (csrc/gemm_f32s.hip) at the two shapes the step runs it on most (W_a forward /
    # input gradient; the heads' first Linear): same accounting
    # ... plus the two LSTM-layer products it carries since round 4 (input projection, dX), the largest launches of the step
    for (M, Nn, Kk) in ((2 * B * T, d, d), (2 * B * T, 512, 2 * d), (2 * B * T, 4 * d, d), (2 * B * T, d, 4 * d)):
        if M % 256 or Nn % 256 or Kk % 32:
            continue
        Xg = torch.randn(M, Kk, device=dev); Wg = torch.randn(Nn, Kk, device=dev) / Kk ** 0.5; Yg = torch.empty(M, Nn, device=dev)
        name = f"tsg_gemm_f32s[{M}x{Kk} . ({Nn}x{Kk})^T]"
        run(name, lambda: lib.tsg_gemm_f32s(ptr(Xg), ptr(Wg), None, ptr(Yg), M, Nn, Kk, st), (M * Kk + Nn * Kk + M * Nn) * e)
        fl = 3 * 2.0 * M * Nn * Kk
        out[name].update(bound="mfma", mfma_flops=fl, achieved_TFLOPs=round(fl / out[name]["mean_us"] / 1e6, 1),
                         mfma_frac=round(fl / out[name]["mean_us"] / 1e6 / MFMA_BF16_PEAK_TFLOPS, 4),
                         per_cycle_note="mfma_frac is against the nominal 2.5 PFLOP/s (2.4 GHz); per GPU-active cycle the committed PMC pass of the step "
                                        "(profiles/r4/mfma_utilisation_f32s_v1.txt, a profile citation) gives 55 % for this kernel and 56 % for the weight gradient")
    return out


def pmc_traffic(kernel, B, T, N, d, launch_B=None, dtype="f32"):
    """(HBM bytes per launch, source file) from the COMMITTED rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE, separate passes;
    the newest round's first) -- a citation of a profile of the same kernel and shape, not something this run observed:
    the JSON says so in `traffic_source`.  (None, None) when no pass exists for this shape and storage dtype."""
    launch_B = launch_B or B
    fname = "k1_pmc_traffic_bf16.json" if dtype == "bf16" else "k1_pmc_traffic.json"
    rounds = sorted((d_ for d_ in os.listdir(os.path.join(ROOT, "profiles")) if re.fullmatch(r"r\d+", d_)), key=lambda d_: -int(d_[1:]))
    for rnd in rounds:                                    # the newest committed pass of this shape and storage dtype
        rel = os.path.join("profiles", rnd, fname)
        try:
            with open(os.path.join(ROOT, rel)) as f:
                j = json.load(f)
            if j["shape"] != {"B": B, "T": T, "N": N, "d": d, "dtype": dtype}:
                continue
            exact = j["kernels"].get(f"{kernel}@B{launch_B}")          # a pass at the launch's own pair count, if committed
            if exact:
                return exact["hbm_bytes_per_launch"], rel
            if kernel in j["kernels"]:
                return int(j["kernels"][kernel]["hbm_bytes_per_launch"] * launch_B / B), rel + f" (scaled from B={B})"
        except (OSError, KeyError, ValueError):
            pass
    return None, None


def cpu_baseline(kind, params, T, N, sample_B):
    """The CPU oracle on the same step, bounded sample (about 10-30 s of CPU work)."""
    from oracle import tsg_oracle as O
    torch.manual_seed(0)
    model = engine.build_model(kind, params)
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in model.state_dict().items()}

    def step(B):
        b = data.synthetic_batch(B, T, N, pair=(kind == "gmd"))
        if kind == "gmd":
            g, pg = b["gt"], b["pseudo_gt"]
            out = O.gmd_forward(sd, b["query"], b["video"], b["video_mask"], b["pseudo_video"], b["video_mask"],
                                g["temporal_labels"], g["fore_masks"], g["back_masks"],
                                pg["temporal_labels"], pg["fore_masks"], pg["back_masks"])
            loss, _ = O.gmd_losses(out, b["video_mask"], b["video_mask"], g, pg)
        else:
            out = O.baseline_forward(sd, b["video"], b["query"], b["video_mask"])
            loss = O.span_ground_loss(out["start"], out["end"], b["gt"]["framestps"])
        loss.backward()
    # The oracle's LSTM is a Python loop of small GEMMs: beyond ~16 intra-op threads the fork / join overhead dominates, so the thread
    # count is capped at the fastest setting and stated (tools/oracle_threads_probe.py on the GPU box's 256 host threads, a step of 16
    # pairs: 8 threads 5.3 pairs/s, 16: 6.6, 32: 3.8, 64: 1.9; torch's default of 128: minutes per step).
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    sB = max(1, min(16, sample_B))                         # pairs per CPU step
    log(f"cpu baseline: oracle {kind} steps of {sB} pairs on {cores} of {os.cpu_count()} host threads")
    step(2)                                   # warm-up
    t0 = time.time()
    done = 0
    while done < sample_B and time.time() - t0 < 30.0:     # bounded: about 10-30 s of CPU work
        step(sB)
        done += sB
    dt = time.time() - t0
    log(f"cpu baseline: {done} pairs in {dt:.1f} s")
    return {"value": round(done / dt, 3), "unit": "pairs/s", "cores": cores, "kind": "port",
            "sample": f"{done // sB} {kind.upper()} train steps (fwd+losses+bwd, fp32) of {sB} pairs at T={T},N={N},"
                      f"d={2 * params['video_rnn_hiddendim']} by oracle/tsg_oracle.py on {cores} of {os.cpu_count()} host threads, {dt:.1f} s"}


def main():
    a = parse_args()

    rank = int(os.environ.get("RANK", 0)); world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}, "
                         f"or without a launcher environment (bench.py then starts its own ranks)")
    if torch.cuda.device_count() < max(1, local + 1):          # counting devices does not initialise the GPU
        raise SystemExit(f"bench.py needs an MI355X per rank: {torch.cuda.device_count()} visible, local rank {local} "
                         "(the hot path has no CPU fallback)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = world > 1 or os.environ.get("TSG_FORCE_DIST") == "1"      # the latter: exercise the RCCL path on one GPU
    if use_dist:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        if dist.get_world_size() != a.gpus:
            raise SystemExit(f"RCCL world size {dist.get_world_size()} != --gpus {a.gpus}")

    params = engine.default_params(video_rnn_hiddendim=a.d // 2, sent_rnn_hiddendim=a.d // 2,
                                   video_len=a.T, sent_len=a.N, predictor=a.predictor)
    torch.manual_seed(0)
    log(f"building {a.model} (d={a.d}) on {torch.cuda.get_device_name(local)}")
    model = engine.build_model(a.model, params).to(dev).train()
    # gradient exchange: bucket all-reduces launched DURING the backward, but never beside a persistent LSTM kernel (those need every
    # CU for the length of a layer): dp's gated mode launches the complete buckets right BEHIND each persistent launch of the backward --
    # RCCL's stream picks them up when that kernel has finished and they run under the dX / weight-gradient GEMMs that follow -- and the
    # hot path fences them in front of the next persistent launch (functional.set_persistent_gate).  TSG_DP_OVERLAP=0: one all-reduce of
    # the whole buffer after the backward (rounds 1-4).  The graph-replay leg below always exchanges after the backward.
    overlap = os.environ.get("TSG_DP_OVERLAP", "1") == "1"
    dp = FlatGradAllReduce(model, overlap=overlap, gated=True, bucket_mb=float(os.environ.get("TSG_DP_BUCKET_MB", "24")))
    if dp.gated:
        functional.set_persistent_gate(dp)
    opt = engine.make_optimizer(model, params)
    # weak scaling: B pairs per rank.  strong scaling: B is the GLOBAL batch; every rank builds the same global batch (same seed)
    # and keeps its contiguous shard, as the reference's DataParallel scatters one batch over the GPUs (train.py:343)
    if a.scaling == "strong":
        if a.B % world:
            raise SystemExit(f"--scaling strong: the global batch {a.B} must be divisible by --gpus {world}")
        from shufflingvideosfortsg_amd.dp import shard_batch
        Bl, Bglobal = a.B // world, a.B
        batch = shard_batch(data.synthetic_batch(a.B, a.T, a.N, seed=1234, pair=(a.model == "gmd")), rank, world)
        batch = data.to_device(batch, dev)
    else:
        Bl, Bglobal = a.B, a.B * world
        batch = data.synthetic_batch(a.B, a.T, a.N, seed=1234 + rank, pair=(a.model == "gmd"), device=dev)
    # the bf16 storage mode keeps the clip features resident in its storage dtype (inputs are resident in HBM before the timed region
    # in every mode; which dtype they are resident in is part of the mode)
    batch_bf16 = dict(batch)
    if "pseudo_video" in batch:                              # (the two streams stay back to back: the model's batch concatenation is a view)
        vb = data.adjacent_cat(batch["video"], batch["pseudo_video"]).to(torch.bfloat16)
        batch_bf16["video"], batch_bf16["pseudo_video"] = vb[:batch["video"].shape[0]], vb[batch["video"].shape[0]:]
    else:
        batch_bf16["video"] = batch["video"].to(torch.bfloat16)
    MODES = {"f32": None, "bf16": "bf16", "f32s": "f32s", "bf16g": torch.bfloat16}
    gdt = MODES[a.dtype]

    def forward():
        bt = batch_bf16 if (isinstance(gdt, str) and gdt == "bf16") else batch
        with engine.precision(gdt):
            if a.model == "gmd":
                return engine.gmd_step(model, bt, params)[0]
            return engine.baseline_step(model, bt)[0]

    def train_step():
        dp.zero_grad()
        loss = forward()
        loss.backward()                           # runs in the forward's mode (saved in the autograd contexts)
        guard = engine.step_guard(loss)           # this rank's skip flag (non-finite loss / expired bounded wait), on the device
        dp.finish(guard=guard)                    # ... reduced over the ranks inside the gradient all-reduce
        engine.optimizer_step(opt, loss, dp=dp, guard=guard)   # the update is skipped ON THE DEVICE, on every rank alike (no sync)
        return loss

    def fwd_step():
        with torch.no_grad():
            return forward()

    step = fwd_step if a.fwd_only else train_step

    def timed(fn, n):
        """n calls bracketed by barrier + synchronize on both sides -> (max-over-ranks seconds, host enqueue seconds, last loss)"""
        import gc
        gc.collect()
        gc.disable()                                      # no cyclic-GC pauses of the host inside a timed region
        if use_dist:
            dist.barrier()
        engine.skipped_updates(reset=True)                # (synchronises: reads the device counter)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            last = fn()
        t_enq = time.perf_counter() - t0
        gc.enable()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        # updates the guard skipped inside this region (non-finite loss / gradient, expired bounded wait -- on ANY rank: the flag is reduced with
        # the gradients): steps that did not update are not training steps, so a region that skipped reports no throughput (round-5 review)
        timed.skipped = engine.skipped_updates()
        if timed.skipped:
            raise SystemExit(f"{timed.skipped} optimizer update(s) were skipped by the step guard inside a timed region of {n} steps: no throughput reported")
        return dt, t_enq, last

    if a.graph_only:
        opt_g = engine.make_optimizer(model, params, capturable=True)
        dp.set_overlap(False)
        functional.set_persistent_gate(None)
        gstep = engine.GraphedTrainStep(model, opt_g, lambda m, b: forward(), batch, dp=dp)
        for _ in range(3):
            gstep()
        dt4, enq4, loss4 = timed(gstep, a.steps)
        functional.check_lstm_errors()
        out = {"graph_replay": {"value": round(Bglobal * a.steps / dt4, 2), "unit": "pairs/s", "ms_per_step": round(dt4 / a.steps * 1e3, 3),
                                "host_enqueue_ms_per_step": round(enq4 / a.steps * 1e3, 3), "finite": bool(torch.isfinite(loss4)),
                                "skipped_updates": timed.skipped,
                                "note": "same step and mode replayed from two HIP graphs (forward+losses+backward | guarded Adam), gradient "
                                        "exchange eager between them (engine.GraphedTrainStep); measured in a child process"}}
        os.write(_RESULT_FD, (json.dumps(out) + "\n").encode())
        return

    log("batch resident; warm-up")
    for i in range(a.warmup):
        step()
        torch.cuda.synchronize()
        log(f"warm-up step {i} done")
    # event pairs around the hot-path kernel launches (K1 / K1g / K3; K2 when it is in the step) -- not around the LSTM and
    # operand-split launches, whose barrier packets would cost milliseconds per step (rocprof has their times)
    functional.kernel_timer.enable(only=None if a.time_all else ("tsg_scdm", "tsg_boundary", "tsg_mha", "tsg_match_head", "tsg_moment_pool"))
    dt, t_enq, loss = timed(step, a.steps)                # t_enq: host time to enqueue the K steps (== dt when the host is the limit)
    skipped_main = timed.skipped
    functional.kernel_timer.disable()
    ms_ = torch.cuda.memory_stats()
    log(f"timed {a.steps} steps in {dt:.3f} s (host enqueue {t_enq:.3f} s); device allocs {ms_.get('num_device_alloc')} frees {ms_.get('num_device_free')} "
        f"reserved {ms_.get('reserved_bytes.all.current', 0) / 2**30:.1f} GiB")
    if not torch.isfinite(loss):
        raise SystemExit("non-finite loss in the timed region")
    functional.check_lstm_errors()                        # a persistent LSTM launch whose bounded wait expired -> invalid run

    # side measurements, outside the timed region above and never `value`: the same K steps in the other library-GEMM
    # modes (strict fp32 library GEMMs, and the bf16 operands that BASELINE config 2 names)
    NOTES = {"f32": "all-f32 (rocBLAS fp32 MFMA GEMMs)",
             "bf16": "bf16 STORAGE (BASELINE configs 2 / 4): activations and their gradients in HBM as bf16 through every hand-written "
                     "kernel of the step (dtype TSG_BF16: K1g, K3, K5, persistent LSTM), plain bf16 MFMA GEMMs, fp32 arithmetic inside "
                     "the kernels, fp32 cell state / softmax / losses / master weights / optimizer; ~1e-2 relative to the fp32 reference",
             "bf16g": "library GEMM operands bf16, fp32 accumulate, fp32 storage; HIP kernels, recurrent state, softmax, losses, "
                      "optimizer f32; deviates ~1e-3 from the fp32 reference",
             "f32s": "LSTM GEMMs, LSTM recurrence products and the large projections as split-precision bf16 MFMA products "
                     "(hi*hi+hi*lo+lo*hi, fp32 accumulate: <= 2e-5 x scale against float64 -- 2^-16 product error, ~100x a true fp32 GEMM's rounding; passes the 1e-4 "
                     "output tolerance of the parity suite); everything else f32"}
    alt = []
    gdt_main = gdt
    for mode in ([] if a.no_alt else [m for m in MODES if m != a.dtype]):
        gdt = MODES[mode]
        try:                                              # (a side measurement must not take the headline down with it)
            for _ in range(2):
                step()
            dt2, _, loss2 = timed(step, a.steps)
            alt.append({"dtype": mode, "value": round(Bglobal * a.steps / dt2, 2), "unit": "pairs/s",
                        "ms_per_step": round(dt2 / a.steps * 1e3, 3), "finite": bool(torch.isfinite(loss2)), "note": NOTES[mode]})
            log(f"alternate GEMM mode {mode}: {alt[-1]['ms_per_step']} ms/step")
            del loss2
        except Exception as e:                            # noqa: BLE001
            alt.append({"dtype": mode, "error": f"{type(e).__name__}: {e}"[:300], "note": NOTES[mode]})
            log(f"alternate GEMM mode {mode} failed: {alt[-1]['error']}")
    gdt = gdt_main
    functional.set_gemm_dtype(gdt_main)
    fwd_only = None
    if not a.fwd_only and not a.no_alt:                   # forward-only pairs/s, reported separately
        for _ in range(max(a.warmup, 2)):                 # the first no_grad passes after training steps run host-bound (allocator
            fwd_step()                                    # re-fit: 8.7 instead of 3.5 ms of host time per step); not part of the rate
        torch.cuda.synchronize()
        dt3, enq3, loss3 = timed(fwd_step, a.steps)
        ms_ = torch.cuda.memory_stats()
        log(f"forward-only: host enqueue {enq3 / a.steps * 1e3:.3f} ms/step; device allocs {ms_.get('num_device_alloc')} frees {ms_.get('num_device_free')} "
            f"reserved {ms_.get('reserved_bytes.all.current', 0) / 2**30:.1f} GiB retries {ms_.get('num_alloc_retries')}")
        fwd_only = {"value": round(Bglobal * a.steps / dt3, 2), "unit": "pairs/s", "ms_per_step": round(dt3 / a.steps * 1e3, 3),
                    "finite": bool(torch.isfinite(loss3)), "note": "forward + losses under no_grad, same batch and mode as `value`"}
        log(f"forward-only: {fwd_only['ms_per_step']} ms/step")
        del loss3
    functional.check_lstm_errors()
    # the same train step replayed from two HIP graphs IN THIS PROCESS (every rank), gradient exchange eager between them:
    # what --gpus N > 1 runs by default.  A rank that cannot capture reports it and ALL ranks skip the leg (agreed by an
    # all-reduce) -- a half-captured world would deadlock in the exchange.
    graph_inproc = None
    graph_replay_bf16 = None
    enq_ranks = [round(t_enq / a.steps * 1e3, 3)]
    if use_dist:
        t = torch.zeros(world, device=dev, dtype=torch.float64)
        t[rank] = t_enq / a.steps * 1e3
        dist.all_reduce(t)
        enq_ranks = [round(float(v), 3) for v in t.tolist()]
    want_graph = (not a.fwd_only) and (a.graph == "on" or (a.graph == "auto" and (world > 1 or not a.no_alt)))
    if want_graph:
        gstep, err = None, ""
        # nothing of the eager steps may stay alive: their autograd graph owns AccumulateGrad nodes bound to the eager stream, and the
        # capture runs on GraphedTrainStep's own stream (a stale node there breaks the capture)
        loss = float(loss)
        dp.zero_grad()
        for p_ in model.parameters():
            p_.grad = None
        import gc
        gc.collect()
        torch.cuda.synchronize()
        try:
            opt_g = engine.make_optimizer(model, params, capturable=True)
            dp.set_overlap(False)                                 # graph replay: ONE exchange of the static gradients behind graph A
            functional.set_persistent_gate(None)
            gstep = engine.GraphedTrainStep(model, opt_g, lambda m, b: forward(), batch, dp=dp)
        except Exception as e:                                    # noqa: BLE001
            err = f"{type(e).__name__}: {e}"[:300]
        ok = torch.tensor([1.0 if gstep is not None else 0.0], device=dev)
        if use_dist:
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if float(ok.item()) > 0:
            for _ in range(3):
                gstep()
            dtg, enqg, lossg = timed(gstep, a.steps)
            functional.check_lstm_errors()
            genq = [round(enqg / a.steps * 1e3, 3)]
            if use_dist:
                t = torch.zeros(world, device=dev, dtype=torch.float64)
                t[rank] = enqg / a.steps * 1e3
                dist.all_reduce(t)
                genq = [round(float(v), 3) for v in t.tolist()]
            graph_inproc = {"value": round(Bglobal * a.steps / dtg, 2), "unit": "pairs/s", "ms_per_step": round(dtg / a.steps * 1e3, 3),
                            "host_enqueue_ms_per_step_by_rank": genq, "finite": bool(torch.isfinite(lossg)), "skipped_updates": timed.skipped,
                            "note": "same step, mode and batch replayed from two HIP graphs in this process on every rank "
                                    "(forward+losses+backward | guarded Adam), RCCL gradient exchange eager between them"}
            log(f"graph replay (in process): {graph_inproc['ms_per_step']} ms/step")
        else:
            graph_inproc = {"error": err or "another rank could not capture the step"}
            log(f"graph replay (in process) skipped: {graph_inproc['error']}")

    graph_replay = None
    if not a.fwd_only and not a.no_alt and world == 1:
        # the same measurement in a CHILD process started from this one (a fresh capture state; a failure there cannot take this result
        # down).  One GPU only: the child shares it.  (a) fallback for `value` when the in-process graph leg above did not produce a number;
        # (b) ALWAYS for the bf16 storage mode -- BASELINE config 2's own dtype -- as its own graph-replayed line next to `value`
        # (round-5 review item 3): eager, that mode is bound by the host's enqueue time, not by the GPU.
        import subprocess

        def graph_child(dtype):
            cmd = [sys.executable, os.path.abspath(__file__), "--graph-only", "--steps", str(a.steps), "--model", a.model, "--B", str(Bl),
                   "--T", str(a.T), "--N", str(a.N), "--d", str(a.d), "--dtype", dtype, "--predictor", a.predictor]
            try:
                r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600,
                                   env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
                lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"graph_replay"')]
                return json.loads(lines[-1])["graph_replay"] if lines else {"error": f"child exit code {r.returncode}: {r.stderr[-300:]}"}
            except Exception as e:                                # noqa: BLE001
                return {"error": f"{type(e).__name__}: {e}"[:300]}
        if not (graph_inproc and "value" in graph_inproc):
            graph_replay = graph_child(a.dtype)
            log(f"graph replay: {graph_replay}")
        if a.dtype != "bf16":
            graph_replay_bf16 = graph_child("bf16")
            log(f"graph replay, bf16 storage: {graph_replay_bf16}")
    if rank == 0:
        kt = functional.kernel_timer.summary()          # (name, dims) -> (mean us, launches, median us)
        ks = functional.kernel_timer.kernel_summary()   # the K1 launches bracketed by their OWN event pair (tsg_time_next_launch)
        kern = {}
        for (name, dims), (us, n, med) in sorted(kt.items()):
            entry = {"mean_us": round(us, 2), "median_us": round(med, 2), "launches": n, "dims": list(dims)}
            if (name, dims) in ks:                        # the kernel alone (what rocprofv3's kernel trace reports); the pair recorded around
                kus, kn, kmed = ks[(name, dims)]          # the call also holds the dispatch gap in front of the kernel (~3 us)
                entry.update(mean_us=round(kus, 2), median_us=round(kmed, 2), launches=kn, around_call_mean_us=round(us, 2),
                             timing="HIP event pair of the launch itself (hipExtLaunchKernel start / stop events on the launch stream)")
                us = kus
            key = {"tsg_scdm_attn_fwd": "scdm_fwd", "tsg_scdm_attn_bwd": "scdm_bwd",
                   "tsg_scdm_gate_fwd": "scdm_gate_fwd", "tsg_scdm_gate_bwd": "scdm_gate_bwd",
                   "tsg_boundary_score_fwd": "boundary_fwd", "tsg_boundary_score_bwd": "boundary_bwd",
                   "tsg_boundary_score_bwd_ws": "boundary_bwd"}.get(name)
            if name == "tsg_boundary_score_bwd_ws":       # (ws_bytes, B, T, Hm, dtype): the workspace size is not a shape
                dims = dims[1:]
                entry["dims"] = list(dims)
            if name == "tsg_match_head_gemm":             # (ldx, ldw, ws_bytes, M, T, N, K, act): K5 as the epilogue of its own GEMM -- MFMA work
                Mq, Tq, Nq, Kq = dims[-5:-1]
                fl = 2.0 * Mq * Nq * Kq * 3               # three bf16 products per fp32 product (hi*hi + hi*lo + lo*hi)
                entry.update(dims=[Mq, Tq, Nq, Kq], bound="mfma", mfma_flops=fl, achieved_TFLOPs=round(fl / us / 1e6, 1),
                             frac=round(fl / us / 1e6 / MFMA_BF16_PEAK_TFLOPS, 4),
                             note="matching head fused into its first-Linear GEMM (split-precision, 256-row tiles, 4 column tiles per row tile "
                                  "meeting at a ticket): GEMM + relu + w2-dot; y is written for the backward in training")
                kern[f"{name}{[Mq, Tq, Nq, Kq]}"] = entry
                continue
            if name in ("tsg_moment_pool_fwd", "tsg_moment_pool_bwd"):      # (B, T, D, dtype): one pass over feat / dfeat
                Bq, Tq, Dq, dtq = dims[-4:]
                by = Bq * Tq * Dq * (2 if dtq == 1 else 4)
                entry.update(dims=[Bq, Tq, Dq], alg_bytes=by, alg_bytes_formula="B*T*D*e  [one pass over the clip features]", achieved_GBs=round(by / us / 1e3, 1),
                             frac=round(by / us / 1e3 / HBM_PEAK_GBS, 4))
                kern[f"{name}{[Bq, Tq, Dq]}"] = entry
                continue
            if name == "tsg_boundary_head_gemm":          # (ldx, ldw, ws_bytes, B, T, Hm, K): K3 as the epilogue of its own GEMM -- MFMA work
                Bq, Tq, Hmq, Kq = dims[-4:]
                fl = 2.0 * Bq * Tq * 2 * Hmq * Kq * 3     # three bf16 products per fp32 product (hi*hi + hi*lo + lo*hi)
                entry.update(dims=[Bq, Tq, Hmq, Kq], bound="mfma", mfma_flops=fl, achieved_TFLOPs=round(fl / us / 1e6, 1),
                             frac=round(fl / us / 1e6 / MFMA_BF16_PEAK_TFLOPS, 4),
                             note="boundary head fused into its first-Linear GEMM (split-precision, 64/128/256-row tiles): GEMM + tanh + "
                                  "w2-dot + masked softmax; y is written for the backward in training")
                kern[f"{name}{[Bq, Tq, Hmq, Kq]}"] = entry
                continue
            esz = 2 if (dims and dims[-1] == 1) else 4    # last integer argument = dtype: TSG_BF16 (1) stores activations in 2 bytes
            if key and key.startswith("scdm"):            # dims = (B, T, N, H, Ds, dtype)
                by = alg_bytes(key, dims[0], dims[1], dims[2], dims[3], e=esz)
            elif key:                                     # dims = (B, T, Hm, dtype)
                by = alg_bytes(key, dims[0], dims[1], 0, 0, Hm=dims[2], e=esz)
            else:
                by = None
            if by:
                entry.update(alg_bytes=by, alg_bytes_formula=ALG_FORMULA[key] + f", e={esz}", achieved_GBs=round(by / us / 1e3, 1),
                             frac=round(by / us / 1e3 / HBM_PEAK_GBS, 4))
            kern[f"{name}{list(dims[:-1])}"] = entry
        if not a.no_micro:
            log("stand-alone K1 / K2 launches")
            kern.update(micro_kernels(Bl, a.T, a.N, a.d))
        # the cross-attention kernel as the train step launches it: scdm_fwd_kernel with the gate epilogue
        # (tsg_scdm_gate_fwd) when the recalibration layer is fused, else the plain tsg_scdm_attn_fwd
        cands = [(k, v) for k, v in kern.items() if k.startswith(("tsg_scdm_gate_fwd[", "tsg_scdm_attn_fwd[")) and "dims" in v]
        k1name, k1 = max(cands, key=lambda kv: kv[1]["launches"], default=("", {}))
        gate = k1name.startswith("tsg_scdm_gate_fwd")
        k1B = (k1.get("dims") or [Bl])[0]
        k1bf = bool(k1.get("dims")) and k1["dims"][-1] == 1
        tr, tr_src = pmc_traffic("scdm_fwd_kernel[gate]" if gate else "scdm_fwd_kernel", 64, a.T, a.N, a.d, launch_B=k1B,
                                 dtype="bf16" if k1bf else "f32")                               # PMC passes at B=64 (+ exact ones)
        k1split = bool(k1.get("dims")) and k1["dims"][-1] in (1, 2)     # TSG_F32S / TSG_BF16: the role-specialised matrix-pipe kernel (H = Ds = 256 CT)
        roof = {"kernel": "%s<GATE=%s> (%s)" % ("scdm_fwd_ws_kernel" if k1split and a.d in (256, 512, 1024) else "scdm_fwd_kernel",
                                                "true" if gate else "false", k1name.split("[")[0]), "bound": "hbm",
                "achieved": k1.get("achieved_GBs"), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": k1.get("frac"), "traffic": tr,
                "traffic_source": (tr_src + ": committed rocprofv3 PMC pass (FETCH_SIZE x2 + WRITE_SIZE) of the same kernel and shape -- a "
                                   "profile citation, not observed by this run") if tr_src else None,
                "pairs_per_launch": k1B,
                "alg_bytes_per_launch": k1.get("alg_bytes"), "alg_bytes_formula": k1.get("alg_bytes_formula"),
                "mean_launch_us": k1.get("mean_us"), "launches_timed": k1.get("launches"),
                "timing": k1.get("timing", "HIP event pair recorded around the call on the launch stream"),
                "around_call_mean_us": k1.get("around_call_mean_us"),
                # round-3 review: state the kernel's own ceiling instead of chasing the last 10 %
                "ceiling_note": ("clock-bound, not issue- or bandwidth-bound (DESIGN.md section 4.1): PMC traffic is 1.02 x algorithmic; on one box the same "
                                 "launch takes 62 us on random operands and 53 us on all-zero operands, and the in-kernel clock reads 1.6-1.7 GHz of the "
                                 "nominal 2.4 (profiles/r6/k1g_fwd_power_bound_probe_v1.txt): the VALU-dense score loop (3.75 VALU instructions and 4 LDS "
                                 "bytes per (t,n,k) element, one v_rcp_f32 per four elements) makes the chip hold its clock down, so instruction savings "
                                 "return as a lower clock.  bf16 storage: the SAME fp32 score loop on half the bytes -- the launch takes the same time, "
                                 "so its fraction of the HBM roofline halves")
                if gate else None}
        # one line per run in the log, so that a regression of the roofline kernel is visible from round to round (round-5 review)
        log(f"roofline kernel {roof['kernel']}: {roof['mean_launch_us']} us per {k1B}-pair launch in the step (its own event pair; "
            f"{roof.get('around_call_mean_us')} us around the call) = {roof['frac']} of {HBM_PEAK_GBS / 1e3:.0f} TB/s")
        what = "fwd-only" if a.fwd_only else "fwd+bwd"
        wl = (f"{a.model}_forward: fwd+losses under no_grad, " if a.fwd_only else
              f"{a.model}_train_step: fwd+losses+bwd+grad-allreduce+Adam, ")
        eager = {"value": round(Bglobal * a.steps / dt, 2), "ms_per_step": round(dt / a.steps * 1e3, 3)}
        value, ms_step, value_mode = eager["value"], eager["ms_per_step"], "eager"
        if graph_inproc and "value" in graph_inproc and graph_inproc["finite"] and graph_inproc["value"] > value:
            value, ms_step, value_mode = graph_inproc["value"], graph_inproc["ms_per_step"], "graph_replay"
        bdesc = (f"global B={a.B} sharded over {world} GPU(s) = {Bl}/GPU (strong scaling)" if a.scaling == "strong" else f"B={a.B}/GPU")
        out = {"metric": f"clip-query pairs/sec {what} at B={a.B},T={a.T},d={a.d}", "value": value,
               "unit": "pairs/s", "n_gpus": world, "rccl_ranks": dist.get_world_size() if use_dist else 1,
               "steps": a.steps, "warmup": a.warmup,
               "ms_per_step": ms_step, "higher_is_better": True, "scaling": a.scaling,
               "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
               "value_mode": value_mode + (": the same full step replayed from two HIP graphs per rank (engine.GraphedTrainStep)"
                                           if value_mode == "graph_replay" else ": one Python-enqueued step per iteration"),
               "eager": eager, "graph_replay_in_process": graph_inproc, "skipped_updates": skipped_main,
               # the same step in the two other arithmetic modes, as top-level fields (round-5 review items 3 and 7): strict fp32 (rocBLAS fp32
               # GEMMs, every kernel fp32) and bf16 storage (BASELINE config 2's dtype) -- the better of its eager and graph-replayed legs
               "value_f32": next((m["value"] for m in alt if m["dtype"] == "f32" and "value" in m), None),
               "value_bf16": max([m["value"] for m in alt if m["dtype"] == "bf16" and "value" in m] +
                                 ([graph_replay_bf16["value"]] if graph_replay_bf16 and graph_replay_bf16.get("finite") else []), default=None),
               "graph_replay_bf16": graph_replay_bf16,
               "config": {"workload": wl + f"{bdesc},T_clip={a.T},T_word={a.N},d={a.d}, i3d width 1024, GloVe 300"
                                      + ("" if a.predictor == "mlp" else f", boundary head {a.predictor}")
                                      + ("" if a.dtype == "f32" else "; " + NOTES[a.dtype]),
                          "global_batch": Bglobal, "pairs_per_gpu": Bl, "parallelism": f"dp{world}", "grad_bytes": dp.grad_bytes},
               "roofline": roof, "alt_gemm_modes": alt, "fwd_only": fwd_only, "graph_replay": graph_replay, "kernels": kern,
               "host_enqueue_ms_per_step": round(t_enq / a.steps * 1e3, 3), "host_enqueue_ms_per_step_by_rank": enq_ranks,
               "cpu_baseline": cpu_baseline(a.model, params, a.T, a.N, a.cpu_sample) if (a.cpu_sample > 0 and world == 1) else None}
        os.write(_RESULT_FD, (json.dumps(out) + "\n").encode())
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
