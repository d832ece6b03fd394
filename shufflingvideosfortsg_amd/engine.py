"""Model construction and the train / eval step of the grounding path, with the step semantics of the
reference drivers (grounding/train_baseline.py, train.py, test_baseline.py) but none of their CLI,
logging or checkpoint-directory plumbing.

  * ``default_params`` / ``make_settings`` reproduce the argparse defaults and the four setting dicts
    the reference packs for its model constructors (train.py:50-91).
  * ``baseline_step`` / ``gmd_step``: forward -> losses (train_baseline.py:120-134, train.py:142-165)
    -> backward; the optimiser is ``Adam(lr 1e-3, weight_decay 1e-4, eps 1e-6)`` (train.py:367-384).
  * ``evaluate``: forward under no_grad -> ``span_pred`` -> the reference's submits-JSON schema
    (test_baseline.py:86-137), scored by ``IoU_eval.retrieval_eval``.
"""
from __future__ import annotations

import copy
import logging
from typing import Dict, Iterable

import torch

from . import loss as L
from .model import GMD, Baseline
from .model.networks.attention import masked_softmax

LOG = logging.getLogger("tsg")


def default_params(**over) -> Dict:
    p = dict(video_encoder="query_aware_encoder", video_feature_dim=1024, video_rnn_hiddendim=256,
             video_rnn_layers=2, video_rnn_cell="lstm", mask=False, dropout=0.5, video_len=128,
             sent_encoder="rnn", sent_rnn_hiddendim=256, sent_rnn_layers=2, sent_rnn_cell="lstm", sent_len=20,
             crossmodal="vs", predictor="mlp", span_hidden_dim=128, mlp_hidden_dim=256,
             m_cross="concat", m_temp="none", m_pred="mlp", m_pred_activ="relu", m_pred_hidden=1024,
             loss_m1_lambda=1.0, loss_m2_lambda=1.0, loss_disc_lambda=1.0, lr=1e-3, weight_decay=1e-4)
    p.update(over)
    return p


def make_settings(params: Dict):
    """The four constructor dicts (train.py:50-91)."""
    video = dict(name=params["video_encoder"], input_dim=params["video_feature_dim"],
                 rnn_hidden_dim=params["video_rnn_hiddendim"], rnn_layers=params["video_rnn_layers"],
                 rnn_cell=params["video_rnn_cell"], mask=params["mask"], drop_out=params["dropout"],
                 T=params["video_len"], nblocks=2)
    sent = dict(name=params["sent_encoder"], input_dim=300, rnn_hidden_dim=params["sent_rnn_hiddendim"],
                rnn_layers=params["sent_rnn_layers"], rnn_cell=params["sent_rnn_cell"], drop_out=params["dropout"])
    ground = dict(cross_name=params["crossmodal"], name=params["predictor"],
                  lstm_hidden_dim=params["span_hidden_dim"], mlp_hidden_dim=params["mlp_hidden_dim"])
    if params["predictor"] in ("self_attn", "d"):       # keys SpanPredictor_Boundary reads for this head (SpanPredictor.py:36-40);
        ground.update(attention_nheads=params.get("attention_nheads", 8),      # the reference's drivers never set them
                      position_encoding=params.get("position_encoding", True))
    match = dict(cross=dict(name=params["m_cross"]),
                 temporal=dict(name=params["m_temp"], hidden_dim=256, layers=2, dropout=params["dropout"]),
                 predict=dict(name=params["m_pred"], activation=params["m_pred_activ"], hidden_dim=params["m_pred_hidden"]))
    return video, sent, ground, match


def build_model(kind: str, params: Dict, logger=LOG) -> torch.nn.Module:
    cls = {"qave": Baseline, "baseline": Baseline, "gmd": GMD}[kind.lower()]
    return cls(*copy.deepcopy(make_settings(params)), logger, params["dropout"])


def set_precision(gemm_dtype=None):
    """Set the process-wide arithmetic mode of the path (see ``precision``) and return the previous one.  For code that cannot use
    a ``with`` block (test fixtures with finalizers); everything else uses ``precision``."""
    from . import functional as TF
    prev = TF.get_gemm_dtype()
    TF.set_gemm_dtype(gemm_dtype)
    return prev


class precision:
    """Context manager for the arithmetic mode of the path; the previous mode is RESTORED on exit (round-3 review: it used to set a
    process global and return a null context, so the mode leaked past the block).  ``None`` / fp32 = strict fp32 (parity mode);
    ``"f32s"`` = split-precision bf16 MFMA products, fp32 storage (the headline mode; error <= 2e-5 x scale against float64 -- 2^-16 products, ~100x a true fp32 GEMM's rounding, within the 1e-4 output tolerance); ``"bf16"`` = bf16 STORAGE
    mode (BASELINE configs 2 / 4): activations and their gradients live in HBM as bf16, the hand-written kernels run with dtype
    TSG_BF16 (fp32 arithmetic inside), the GEMMs are plain bf16 MFMA GEMMs, parameters / weight gradients / optimizer state stay
    fp32 -- no autocast involved; ``torch.bfloat16`` = the older operands-only mode: every library GEMM (nn.Linear projections via
    autocast, the LSTM GEMMs via functional.set_gemm_dtype) takes bf16 operands with fp32 accumulation while the HIP kernels and
    all storage stay fp32.  A backward runs in the mode of ITS forward (saved in the autograd context), so ``loss.backward()`` may
    sit outside the block."""

    def __init__(self, gemm_dtype=None):
        self.mode = gemm_dtype
        self._prev = None
        self._autocast = None

    def __enter__(self):
        self._prev = set_precision(self.mode)
        if self.mode not in (None, torch.float32, "f32s", "bf16"):
            self._autocast = torch.autocast("cuda", dtype=self.mode)
            self._autocast.__enter__()
        return self

    def __exit__(self, *exc):
        try:
            if self._autocast is not None:
                self._autocast.__exit__(*exc)
        finally:
            self._autocast = None
            set_precision(self._prev)
        return False


class TsgAdam(torch.optim.Optimizer):
    """torch.optim.Adam(lr, weight_decay (L2), eps) -- the optimizer of the reference (train.py:367-371) -- on the hand-written kernel
    (csrc/adam.hip: ``tsg_adam_step``): every parameter of the model in one launch per 64 tensors (two for GMD's 80) instead of the six
    multi_tensor_apply launches of torch's fused Adam, the update count kept (and advanced) on the device, the skip flag of
    ``optimizer_step`` read on the device.  Same state keys as torch's Adam (``exp_avg``, ``exp_avg_sq``, ``step``: ONE shared device
    scalar here), so ``state_dict()`` round-trips; always graph-capturable.  fp32 CUDA parameters only; ``grad_scale`` folds a 1 / world
    into the update when the gradient exchange summed instead of averaging."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, fused=True, capturable=True))
        self.found_inf = None          # device float (non-zero = skip this update): the protocol of torch's fused optimizers (GradScaler)
        self.grad_scale = 1.0
        self._state_words = {}         # (device, param group index) -> [update count (float), ticket]
        self._tables = {}

    def _words(self, dev, gi=0):
        """The update count of param group ``gi`` on ``dev``: the kernel advances it once per call, so every group keeps its own (with one
        word per device the count rose once per GROUP and later groups were bias-corrected at t + 1).  A parameter whose first gradient arrives late
        still starts at its group's t against zero moments -- torch keeps a count per parameter; the reference's optimizer has one group and every
        parameter receives a gradient in every step."""
        w = self._state_words.get((dev, gi))
        if w is None:
            w = self._state_words[(dev, gi)] = torch.zeros(2, device=dev, dtype=torch.float32)
        return w

    @torch.no_grad()
    def step(self, closure=None):
        import ctypes
        from . import functional as TF
        from ._lib import load, ptr
        loss = closure() if closure is not None else None
        lib = load()
        if self.found_inf is not None:                     # a guarded update also looks at the gradients themselves: a NaN or an infinity
            gs = [p.grad for g in self.param_groups for p in g["params"] if p.grad is not None and p.grad.is_contiguous()]   # in ANY group skips them all
            if gs:
                TF.check(lib.tsg_grads_nonfinite(len(gs), (ctypes.c_void_p * len(gs))(*[g.data_ptr() for g in gs]),
                                                 (ctypes.c_longlong * len(gs))(*[g.numel() for g in gs]), ptr(self.found_inf),
                                                 TF.stream_of(gs[0])), "tsg_grads_nonfinite")
        for gi, group in enumerate(self.param_groups):
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            dev = ps[0].device
            words = self._words(dev, gi)
            for p in ps:
                if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.device == dev):
                    raise ValueError("TsgAdam: fp32 contiguous CUDA parameters on one device only")
                st = self.state[p]
                if "exp_avg" not in st:
                    st["exp_avg"], st["exp_avg_sq"] = torch.zeros_like(p), torch.zeros_like(p)
                if st.get("step") is None or st["step"].data_ptr() != words.data_ptr():
                    if isinstance(st.get("step"), torch.Tensor) and st["step"].numel() == 1 and st["step"].data_ptr() != words.data_ptr():
                        words[0] = float(st["step"])                      # a loaded state_dict brings its own count: adopt it once
                    st["step"] = words[0:1].view(())
            n = len(ps)
            grads = [p.grad if p.grad.is_contiguous() else p.grad.contiguous() for p in ps]
            arr = lambda ts: (ctypes.c_void_p * n)(*[t.data_ptr() for t in ts])
            numel = (ctypes.c_longlong * n)(*[p.numel() for p in ps])
            b1, b2 = group["betas"]
            skip = self.found_inf
            # bf16 shadows of the parameters the bf16 storage mode has used as GEMM operands (functional.weight_bf16): rewritten by the same kernel
            shadows = [TF.shadow_of(p) for p in ps]
            if any(sh is not None for sh in shadows):
                sh_arr = (ctypes.c_void_p * n)(*[sh.data_ptr() if sh is not None else None for sh in shadows])
                rc = lib.tsg_adam_step_shadow(n, arr(ps), arr(grads), arr([self.state[p]["exp_avg"] for p in ps]), arr([self.state[p]["exp_avg_sq"] for p in ps]),
                                              sh_arr, numel, float(group["lr"]), float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]),
                                              float(self.grad_scale), ptr(words), ptr(skip) if skip is not None else None, TF.stream_of(ps[0]))
            else:
                rc = lib.tsg_adam_step(n, arr(ps), arr(grads), arr([self.state[p]["exp_avg"] for p in ps]), arr([self.state[p]["exp_avg_sq"] for p in ps]),
                                       numel, float(group["lr"]), float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]),
                                       float(self.grad_scale), ptr(words), ptr(skip) if skip is not None else None, TF.stream_of(ps[0]))
            TF.check(rc, "tsg_adam_step")
            for p in ps:                                   # the kernel wrote p (and its shadow) through raw pointers: version counters follow
                TF.restamp_shadow(p)
        return loss


def make_optimizer(model, params, capturable=False):
    """Adam(lr, L2 weight decay, eps=1e-6) as the reference builds it (train.py:367-371); on a GPU the single-pass
    fused implementation of the same update (one kernel over all parameters instead of ~7 foreach passes).
    ``capturable``: step counters on the device, so that the update can be part of a HIP graph (GraphedTrainStep)."""
    ps = list(model.parameters())
    fused = bool(ps) and all(p.is_cuda for p in ps)
    import os
    if fused and os.environ.get("TSG_OWN_ADAM", "1") != "0" and all(p.dtype == torch.float32 for p in ps):
        return TsgAdam(ps, lr=params["lr"], weight_decay=params["weight_decay"], eps=1e-6)      # one launch per 64 tensors (csrc/adam.hip)
    return torch.optim.Adam(ps, lr=params["lr"], weight_decay=params["weight_decay"], eps=1e-6, fused=fused,
                            capturable=bool(capturable and fused))


def step_guard(loss):
    """This rank's skip flag for the step that produced ``loss``, computed ON THE DEVICE after the backward has been enqueued: 1.0
    when the loss is not finite (a persistent LSTM forward whose bounded wait expired leaves NaN sentinels in its output; an
    overflow) OR when any kernel of the step reported an expired bounded wait into the device error word
    (``functional.error_word``: the persistent LSTM backward and the K1 backward's partner exchange run AFTER the loss is known to
    be finite; the K1 backward additionally poisons the affected gradients with NaN).  float32 [1]; no synchronisation."""
    bad = ~torch.isfinite(loss.detach()).reshape(())
    if loss.is_cuda:
        from . import functional as TF
        bad = bad | (TF.error_word(loss.device)[0] != 0)
    return bad.to(torch.float32).reshape(1)


def optimizer_step(opt, loss, dp=None, guard=None):
    """``opt.step()`` guarded on the DEVICE: the fused Adam kernel skips the update -- its ``found_inf`` input, the GradScaler
    mechanism -- when ``step_guard(loss)`` is set on this rank OR, with a gradient exchange ``dp`` (FlatGradAllReduce), on ANY rank:
    the flag is reduced across the ranks inside the gradient all-reduce (``dp.finish(guard=step_guard(loss))`` /
    ``dp.exchange_static(grads, guard)``), because the averaged gradient a healthy rank holds contains the corrupted rank's
    contribution -- with a rank-local guard the replicas would diverge.  So the parameters and the Adam moments are not
    corrupted by a step the host has already enqueued, also when the step is replayed from HIP graphs.  No synchronisation; the
    host-side report follows at the next ``functional.check_kernel_errors()`` (every LSTM call, every graph replay).  With a
    non-fused optimizer (CPU tests) the flag is read on the host instead (one ``.item()``), so the skip semantics are the same there.
    ``guard``: the local flag if the caller has already computed it (it must be the tensor handed to ``dp.finish``)."""
    fused = any(g.get("fused") for g in opt.param_groups)
    bad = step_guard(loss) if guard is None else guard.reshape(1).to(torch.float32)
    if dp is not None and getattr(dp, "active", False) and dp.guard is not None:
        bad = torch.maximum(bad, (dp.guard != 0).to(torch.float32))
    if fused:
        flag = bad.reshape(())
        opt.found_inf = flag
        try:
            opt.step()                                     # (TsgAdam also ORs "a gradient is NaN / inf" into the flag: tsg_grads_nonfinite)
        finally:
            opt.found_inf = None
        _skip_counter(flag.device).add_((flag != 0).to(torch.float32))     # on the device: part of the captured update under graph replay
    elif float(bad.item()) == 0.0:                         # host-side optimizers (CPU tests, non-fused GPU optimizers), with or without a
        opt.step()                                         # gradient exchange: the same rule, the flag read on the host (one sync)
    else:
        _skip_counter(bad.device).add_(1.0)


_SKIPPED = {}              # device -> float32 [] count of optimizer updates the guard has skipped


def _skip_counter(device):
    c = _SKIPPED.get(device)
    if c is None:
        c = _SKIPPED[device] = torch.zeros((), device=device, dtype=torch.float32)
    return c


def skipped_updates(device=None, reset: bool = False) -> int:
    """How many updates ``optimizer_step`` has skipped on ``device`` (default: every device) since the start / the last reset: a
    non-finite loss, an expired bounded wait, a NaN / inf gradient, or -- with a gradient exchange -- any of these on ANY rank.  A
    throughput measured over steps that did not update is not a training throughput: bench.py prints the count and refuses a run
    whose timed region skipped.  Synchronises (reads the device counters)."""
    devs = [d for d in _SKIPPED if device is None or d == torch.device(device)]
    n = int(sum(float(_SKIPPED[d]) for d in devs))
    if reset:
        for d in devs:
            _SKIPPED[d].zero_()
    return n


import os as _os
_GRAPH_PROBE = _os.environ.get("TSG_GRAPH_PROBE", "")     # developer probe of the replay defect: 'ab' / 'ba' host syncs, 'skipb'


class GraphedTrainStep:
    """One training step as two HIP graphs: A = forward + losses + backward, B = the (guarded) Adam update, with the
    gradient exchange of ``dp`` (FlatGradAllReduce; may be inactive) run eagerly between them.  The step's ~390 kernel
    launches -- the C-ABI kernels, their zero-fill nodes, autograd's glue, the fused Adam -- are enqueued by two
    ``hipGraphLaunch`` calls: host time per step drops from ~12 ms to ~0.2 ms (tools/graph_step_probe.py), which is what keeps
    eight ranks on one host from queueing behind their Python threads.  The reference has no counterpart (eager PyTorch 1.x).

    ``step_fn(model, batch) -> loss`` must be capture-safe: static shapes, no host synchronisation, device-resident batch (the
    caller refreshes the batch by copying into the SAME tensors).  ``opt`` must come from ``make_optimizer(capturable=True)``.
    Dropout draws fresh masks on every replay: torch's graph-safe Philox offsets for ``F.dropout``, and for the in-kernel attention
    dropout of K2 a device-resident (seed, offset) pair whose increment is part of the graph (``functional.mha_graph_rng``).
    """

    def __init__(self, model, opt, step_fn, batch, dp=None, warmup=3):
        from . import functional as TF
        from . import _runtime_env
        if not _runtime_env.graph_replay_safe() and _os.environ.get("TSG_GRAPH_UNSAFE") != "1":
            raise RuntimeError("GraphedTrainStep: the ROCm runtime's graph packet capture is on (DEBUG_CLR_GRAPH_PACKET_CAPTURE must be 0 BEFORE the first "
                               "GPU call of the process; import shufflingvideosfortsg_amd before touching the GPU, or export it): replayed graphs can "
                               "read stale data from reused pool blocks (shufflingvideosfortsg_amd/_runtime_env.py).  TSG_GRAPH_UNSAFE=1 overrides.")
        self.model, self.opt, self.dp, self.batch = model, opt, dp, batch
        if dp is not None and dp.overlap:
            raise ValueError("GraphedTrainStep: the gradient exchange must run after the backward (FlatGradAllReduce(overlap=False))")
        self.params = [p for p in model.parameters() if p.requires_grad]
        TF.mha_graph_rng(next(model.parameters()).device)   # K2 dropout state, created outside the capture
        TF.error_word(next(model.parameters()).device)      # the device error word exists (and is registered) before the capture
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                      # eager warm-up on the capture stream: lazy initialisation happens here
            for _ in range(warmup):
                self._zero()
                loss = step_fn(model, batch)
                loss.backward()
                g = step_guard(loss)
                if dp is not None:
                    dp.finish(guard=g)
                optimizer_step(opt, loss, dp=dp, guard=g)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        TF.check_lstm_errors()
        self.graph_a, self.graph_b = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        self._zero()
        # capture on the warm-up stream: autograd's AccumulateGrad nodes were created there, and a capture on another stream
        # would record cross-stream event nodes into the graph (a forked graph: hipGraphLaunch then enqueues node by node)
        # capture_error_mode="thread_local": RCCL's watchdog thread polls the events of the eager exchanges (warm-up steps, dp.adopt below)
        # with hipEventQuery; under the default "global" mode a poll that lands inside a capture aborts the process ("operation not
        # permitted when stream is capturing": seen once in ~6 runs of the world-1 RCCL test).  Nothing of another thread is captured.
        with torch.cuda.graph(self.graph_a, stream=side, capture_error_mode="thread_local"):
            self._zero()
            self.loss = step_fn(model, batch)
            self.loss.backward()
            self.guard = step_guard(self.loss)              # this rank's skip flag: the last node of graph A (after the backward)
        self.grads = [p.grad for p in self.params]          # the graph's static gradient tensors (None: no gradient)
        if dp is not None and dp.active:
            dp.adopt(self.grads, self.guard)                # gather + exchange once: .grad now points into the flat buffer
            torch.cuda.synchronize()                        # the exchange has finished before the second capture begins
        # graph B (a handful of tiny flag tensors) gets a pool of its OWN: nothing it allocates can land in a block graph A still reads or writes
        # (the round-5 stale read involved blocks shared inside a pool; its cause is not established, so no sharing that is not needed)
        with torch.cuda.graph(self.graph_b, stream=side, capture_error_mode="thread_local"):
            optimizer_step(opt, self.loss, dp=dp, guard=self.guard)   # reads the REDUCED flag (dp.guard: static memory of the flat buffer)

    def _zero(self):
        for p in self.params:
            p.grad = None
        if self.dp is not None:
            self.dp.zero_grad()

    def __call__(self):
        """Replay one step on the current batch tensors -> the (static) loss tensor of this step.  The pinned error sink is
        polled on every call (a host read, no synchronisation): no Python LSTM call runs during a replay, so this is where an
        expired wait of an EARLIER replay surfaces (its update was already skipped on the device)."""
        from . import functional as TF
        TF.check_kernel_errors()
        mode = _GRAPH_PROBE
        self.graph_a.replay()
        if "ab" in mode:
            torch.cuda.current_stream().synchronize()
        if self.dp is not None and self.dp.active:
            self.dp.exchange_static(self.grads, self.guard)
        if "skipb" not in mode:
            self.graph_b.replay()
        if "ba" in mode:
            torch.cuda.current_stream().synchronize()
        return self.loss


def baseline_step(model, batch):
    """forward + span_ground_loss of one QAVE batch (train_baseline.py:120-134) -> (loss, span_prob)."""
    out = model(batch["video"], batch["query"], batch["video_mask"], batch["query_mask"])
    return L.span_ground_loss(out["start"], out["end"], batch["gt"]["framestps"]), out


def gmd_step(model, batch, params):
    """forward + the four GMD losses (train.py:131-165) -> (loss, parts, span_prob)."""
    gt, pgt = batch["gt"], batch["pseudo_gt"]
    # the shuffled video's own mask (train.py:146-148 masks the pseudo stream with pseudo_video_mask).  gt_translate keeps
    # nfeats and the moment length (data_augment.py:135-156), so its collate hands over the SAME mask; a batch from another
    # augmentation (crop: aug_nfeats != raw_nfeats, data_augment.py:61,91) carries its own and takes the un-fused losses below
    vm = batch["video_mask"]
    pvm = batch.get("pseudo_video_mask", vm)
    span, om, pm, od, pd = model(batch["query"], batch["query_mask"], batch["video"], vm,
                                 batch["pseudo_video"], pvm,
                                 gt["temporal_labels"], gt["fore_masks"], gt["back_masks"],
                                 pgt["temporal_labels"], pgt["fore_masks"], pgt["back_masks"])
    fs, pfs = gt["framestps"], pgt["framestps"]
    if (om.is_cuda and isinstance(fs, torch.Tensor) and isinstance(pfs, torch.Tensor) and om.dim() == 2 and om.size(1) <= 2048
            and od.dim() == 2 and od.size(1) == 2 and pvm is vm and batch.get("equal_moment_lengths", True)):
        # K4: the four losses in one launch each way (csrc/losses.hip) instead of ~100 small torch launches.  K4 has ONE video
        # mask (shared BCE denominator) and takes the KL slice length from the original span: exact for gt_translate pairs
        # (same mask object, equal moment lengths -- a collate that cannot promise the latter sets equal_moment_lengths=False)
        from . import functional as TF
        lam = (params["loss_m1_lambda"], params["loss_m2_lambda"], params["loss_disc_lambda"])
        parts, total = TF.gmd_losses(span["start"], span["end"], om, pm, od, pd, fs, pfs, gt["temporal_labels"],
                                     pgt["temporal_labels"], batch["video_mask"], lam)
        p = parts.detach()                                  # the parts are for logging; the gradient flows through `total`
        return total, (p[0], lam[0] * p[1], lam[1] * p[2], p[3]), span
    lg = L.span_ground_loss(span["start"], span["end"], gt["framestps"])
    l1 = params["loss_m1_lambda"] * (L.BCE_loss(om, gt["temporal_labels"], vm)
                                     + L.BCE_loss(pm, pgt["temporal_labels"], pvm))
    l2 = params["loss_m2_lambda"] * L.matching_KL_divergence(masked_softmax(om, gt["temporal_labels"]),
                                                             masked_softmax(pm, pgt["temporal_labels"]),
                                                             gt["framestps"], pgt["framestps"])
    ld = L.temporal_order_discrimination_loss(od, pd)
    return lg + l1 + l2 + params["loss_disc_lambda"] * ld, (lg, l1, l2, ld), span


@torch.no_grad()
def evaluate(model, batches: Iterable[Dict], params: Dict = None) -> Dict:
    """-> submits dict {'version','results','external_data','params'} (test_baseline.py:86-137).
    Each batch may carry 'sentences', 'vids', 'durations' lists; predicted 'seconds' are frame indices
    (frame2sec is the identity for raw features, charades.py:275-279)."""
    model.eval()
    sub = {"version": "V0", "results": {}, "external_data": {"used": True, "details": "provided i3D feature"},
           "params": params or {}}
    for bi, batch in enumerate(batches):
        fwd = model.eval_forward if isinstance(model, GMD) else model
        out = fwd(batch["video"], batch["query"], batch["video_mask"], batch["query_mask"])
        pred, score = L.span_pred(out["start"], out["end"])
        pred, score = pred.float().cpu().numpy(), score.cpu().numpy()
        B = pred.shape[0]
        ts = batch["gt"]["timestps"].cpu().numpy()
        for i in range(B):
            vid = batch.get("vids", [f"b{bi}"] * B)[i]
            sub["results"].setdefault(vid, []).append({
                "sentence": batch.get("sentences", [""] * B)[i], "timestamp": pred[i].tolist(),
                "gt_timestamp": ts[i].tolist(), "score": float(score[i]),
                "video_duration": float(batch.get("durations", [0.0] * B)[i])})
    return sub
