"""Data parallelism for the grounding models: one process per GPU, full replica per rank, ONE flat
fp32 gradient buffer, gathered bucket by bucket as the gradients appear and all-reduced (sum) over
RCCL/xGMI in a few large buckets while the backward is still running, then scaled by 1/world.  (The reference only has a single-process
``nn.DataParallel`` wrapper that its launcher pins to one GPU -- train.py:343, helper_function.py:17.)

Clip-query pairs are independent in forward/backward (no BatchNorm; LayerNorm is per row), so the
global batch is split contiguously across ranks and the only exchange is the gradient sum.  Works
with any ``torch.distributed`` backend: "nccl" (= RCCL on ROCm) on the GPUs, "gloo" in the CPU tests.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist


class FlatGradAllReduce:
    """Averages the gradients of ``module`` across ranks through one contiguous buffer.
    Usage per step:  ``dp.zero_grad(); loss.backward(); dp.finish(); optim.step()``.

    Gradients are produced by autograd as individual tensors (``.grad`` starts each step as None, so nothing is
    accumulated into a zeroed buffer: that cost one small add kernel per parameter, ~150 launches a step).  When the
    last parameter of a bucket has its gradient, ONE multi-tensor copy gathers the bucket into its slice of the flat
    buffer, ``.grad`` of its parameters is re-pointed at that slice, and the slice's all-reduce starts -- while the backward
    of the earlier layers is still running.  With one rank there is no exchange and the flat buffer is not touched at all.

    bucket_mb: target bucket size.  xGMI is point-to-point (7 links/GPU), so few large ring
    reductions beat many small ones; 32 MiB buckets give ~6 buckets for the 186 MB d=1024 model.

    Launch order: the buckets are launched in INDEX order on every rank -- bucket b leaves from a hook only when buckets
    0..b-1 have left, ``finish()`` flushes the rest in index order -- so the ranks issue the same sequence of collectives whatever
    subset of parameters received a gradient on each of them (a bucket that completes early on one rank simply waits for the slower
    ranks inside the collective; launching "whichever bucket completed first" mis-paired the all-reduces of ranks with unequal
    gradient presence and hung).

    gated=True (round 5, the GPU default of ``bench.py --gpus N``): the persistent LSTM kernels need every CU for the length of a layer
    (one 512-thread workgroup per CU, all co-resident), so a bucket's all-reduce must never run BESIDE one.  The hot path calls
    ``before_persistent()`` in front of every persistent launch of the backward (all outstanding collectives are waited for ON THE
    STREAM, no host block) and ``after_persistent()`` behind it (every bucket that is complete, in index order, is gathered and
    launched): RCCL's stream picks the collective up when the persistent kernel has finished, and it runs under the dX / weight-gradient
    GEMMs that follow -- ordinary grids that share the CUs -- until the next persistent launch fences it.  Hooks only count.
    """

    def __init__(self, module: torch.nn.Module, process_group: Optional[dist.ProcessGroup] = None,
                 bucket_mb: float = 32.0, overlap: bool = True, broadcast: bool = True, gated: bool = False):
        self.module = module
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.params: List[torch.nn.Parameter] = [p for p in module.parameters() if p.requires_grad]
        if not self.params:
            raise ValueError("module has no trainable parameters")
        dev, dt = self.params[0].device, self.params[0].dtype
        self._force = dist.is_initialized() and __import__("os").environ.get("TSG_FORCE_DIST") == "1"
        self.active = self.world > 1 or self._force          # is there an exchange at all?
        # backward produces gradients roughly in reverse registration order: lay the buffer out that
        # way so that each bucket is a contiguous slice that completes early
        order = list(reversed(self.params))
        self._numel = sum(p.numel() for p in order)
        self._esize = torch.empty(0, dtype=dt).element_size()
        # one extra element behind the gradients: the GUARD slot.  A rank that wants the optimizer update skipped (non-finite loss, an
        # expired bounded wait in one of its kernels) writes 1 there before the exchange; after the (sum / mean) all-reduce the slot is
        # non-zero on EVERY rank, so all replicas skip the same update and stay identical (the guard used to be rank-local
        # while the gradients -- including a corrupted rank's -- were averaged into everyone).  It rides in the gradient all-reduce: no
        # extra collective on the default (exchange-after-backward) and graph-replay protocols.
        self.flat = torch.zeros(self._numel + 1, device=dev, dtype=dt) if self.active else None
        self.buckets = []          # (start, end, [params])
        cap = max(1, int(bucket_mb * (1 << 20) / self._esize))
        off = start = 0
        cur: List[torch.nn.Parameter] = []
        self._bucket_of, self._view = {}, {}
        for p in order:
            n = p.numel()
            if self.active:
                self._view[id(p)] = self.flat[off:off + n].view_as(p)
            self._bucket_of[id(p)] = len(self.buckets)
            off += n; cur.append(p)
            if off - start >= cap:
                self.buckets.append((start, off, cur)); start, cur = off, []
        if cur:
            self.buckets.append((start, off, cur))
        self._ready = [0] * len(self.buckets)
        self._launched = [False] * len(self.buckets)
        self._next = 0             # next bucket to launch: the launch order is the index order on every rank
        self._handles = []
        self._fenced = 0           # handles[:_fenced] have been waited for on the stream
        self.overlap = overlap and self.active
        self._gated_wanted = bool(gated)
        self.gated = self._gated_wanted and self.overlap
        self._hooked = False
        # Parameters that receive no gradient (an unused branch, a frozen head): their bucket would never "complete" and it and ALL later
        # buckets would wait for finish().  After the first exchanged step the ranks agree on the set of parameters that had
        # a gradient on NO rank (one MAX all-reduce of a byte mask); from then on those count as ready at the start of every step.
        self._unused = None        # ids of the parameters assumed to stay without a gradient (None = not learnt yet)
        self._late_grad = False    # a gradient arrived for such a parameter after its bucket had left: this step's update is skipped
        self._gate_calls = 0       # after_persistent() calls of the current backward
        self._gate_seen_prev = True
        self.flushed_in_finish = 0 # buckets (beyond the last one) that finish() had to launch in the last step: lost overlap
        self._warned_flush = False
        # RCCL averages inside the reduction (no extra pass over the buffer); gloo has no AVG: sum, then one division
        self._avg = dist.is_initialized() and dist.get_backend(process_group) == "nccl"
        self._op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
        if self.overlap:
            for p in self.params:
                p.register_post_accumulate_grad_hook(self._hook)
            self._hooked = True
        if broadcast and self.world > 1:
            for t in list(module.parameters()) + list(module.buffers()):
                dist.broadcast(t.data, src=0, group=self.group)
            from . import functional as TF                 # written through .data: the parameters' version counters did not move
            TF.invalidate_shadows(module)

    # -- per-step protocol ----------------------------------------------------------------
    def zero_grad(self):
        for p in self.params:
            p.grad = None
        self._ready = [0] * len(self.buckets)
        if self._unused:
            for b, (_, _, params) in enumerate(self.buckets):
                self._ready[b] = sum(1 for p in params if id(p) in self._unused)
        self._late_grad = False
        # gated mode launches from after_persistent() only: a backward without a persistent kernel (TSG_LSTM_PERSIST=0, a non-HIP LSTM)
        # would exchange everything in finish().  If the previous backward made no gate call, the hooks launch as in the ungated mode.
        self._gate_seen_prev, self._gate_calls = (self._gate_calls > 0 or self._next == 0), 0
        self._launched = [False] * len(self.buckets)
        self._next = 0
        self._handles = []
        self._fenced = 0

    def _gather(self, b):
        """Bucket b's gradients -> its slice of the flat buffer (one multi-tensor copy; parameters without a gradient
        contribute zeros), ``.grad`` re-pointed at the slice."""
        _, _, params = self.buckets[b]
        have = [p for p in params if p.grad is not None]
        none = [self._view[id(p)] for p in params if p.grad is None]
        if have:
            torch._foreach_copy_([self._view[id(p)] for p in have], [p.grad for p in have])
        if none:
            torch._foreach_zero_(none)
        for p in params:
            p.grad = self._view[id(p)]

    def _launch(self, b):
        s, e, _ = self.buckets[b]
        self._gather(b)
        self._launched[b] = True
        self._handles.append(dist.all_reduce(self.flat[s:e], op=self._op, group=self.group, async_op=True))

    def _advance(self, flush: bool = False):
        """Launch, in index order, every bucket whose gradients are all there (``flush``: every remaining bucket; parameters
        without a gradient contribute zeros)."""
        while self._next < len(self.buckets) and (flush or self._ready[self._next] >= len(self.buckets[self._next][2])):
            self._launch(self._next)
            self._next += 1

    def _hook(self, p):
        if not self.overlap:
            return
        b = self._bucket_of[id(p)]
        if self._unused and id(p) in self._unused:         # assumed unused, and here is its gradient
            self._unused.discard(id(p))                    # (counted as ready already)
            if self._launched[b]:                          # its bucket has left without it: the exchanged mean misses this rank's part --
                self._late_grad = True                     # the guard slot makes EVERY rank skip this update; from the next step on the
            return                                         # parameter is waited for again
        self._ready[b] += 1
        if not self.gated or not self._gate_seen_prev:
            self._advance()

    def set_overlap(self, on: bool, gated: Optional[bool] = None):
        """Switch between the bucket-by-bucket exchange during the backward and ONE exchange after it (the graph-replay protocol needs
        the latter).  The hooks are registered on first use and stay; they are inert while the overlap is off."""
        on = bool(on) and self.active
        if on and not self._hooked:
            for p in self.params:
                p.register_post_accumulate_grad_hook(self._hook)
            self._hooked = True
        self.overlap = on
        self.gated = on and (self._gated_wanted if gated is None else bool(gated))

    # -- gating against the persistent kernels (called by functional around every persistent launch of the backward) ------------
    def before_persistent(self):
        """Every collective launched so far must have finished before the kernel enqueued next starts: stream-level waits."""
        for h in self._handles[self._fenced:]:
            h.wait()
        self._fenced = len(self._handles)

    def after_persistent(self):
        """A persistent kernel has just been enqueued: collectives launched now start when it has finished."""
        self._gate_calls += 1
        if self.gated:
            self._advance()

    def _set_guard(self, guard):
        slot = self.flat[self._numel:]
        if guard is None:
            slot.zero_()
        else:
            slot.copy_(guard.detach().reshape(1).to(slot.dtype))

    @property
    def guard(self):
        """After ``finish`` / ``exchange_static``: a 1-element view that is non-zero iff ANY rank passed a non-zero ``guard`` in
        (None when there is no exchange).  ``engine.optimizer_step(..., dp=dp)`` ORs it into the fused optimizer's skip flag."""
        return self.flat[self._numel:] if self.active else None

    def finish(self, guard=None):
        """Complete the gradient exchange: afterwards every rank holds the mean gradient.  ``guard``: this rank's skip flag (a
        1-element tensor, non-zero = skip the update), reduced across the ranks with the gradients (see ``guard``)."""
        if not self.active:
            return
        if self._late_grad:
            guard = torch.ones(1, device=self.flat.device, dtype=self.flat.dtype)
        self._set_guard(guard)
        if self.overlap:
            self.flushed_in_finish = max(0, len(self.buckets) - self._next - 1)     # (the last bucket completes with the backward itself)
            if 2 * self.flushed_in_finish >= len(self.buckets) > 1 and self._unused is not None and not self._warned_flush:   # (a gated backward leaves the tail after its last gate: by design)
                import warnings
                warnings.warn(f"FlatGradAllReduce: {self.flushed_in_finish + 1} of {len(self.buckets)} buckets were still waiting when the backward "
                              "ended (a parameter without a gradient in front of them, or no persistent kernel to gate on): their exchange did not overlap")
                self._warned_flush = True
            learn = self._unused is None and self.world > 1
            if learn:                                      # first exchanged step: which parameters had a gradient on no rank at all?
                mask = torch.tensor([0 if p.grad is None else 1 for p in self.params], device=self.flat.device, dtype=torch.int32)
            self._advance(flush=True)                      # what the hooks / gates have not launched yet, in index order
            if learn:
                dist.all_reduce(mask, op=dist.ReduceOp.MAX, group=self.group)
                self._unused = {id(p) for p, m in zip(self.params, mask.tolist()) if m == 0}
            # the buckets are already in flight: the guard slot goes in a collective of its own (4 bytes)
            self._handles.append(dist.all_reduce(self.flat[self._numel:], op=self._op, group=self.group, async_op=True))
        else:
            for b in range(len(self.buckets)):
                self._gather(b)
            self._handles = [dist.all_reduce(self.flat, op=self._op, group=self.group, async_op=True)]
        for h in self._handles:
            h.wait()
        if not self._avg:
            self.flat.div_(self.world)

    # -- graph-replay protocol (engine.GraphedTrainStep) ---------------------------------------
    def adopt(self, grads, guard=None):
        """Called once after the backward has been captured: ``grads`` are the graph's static gradient tensors (in
        ``self.params`` order).  Gathers and exchanges them once and leaves ``.grad`` of every parameter pointing into the flat
        buffer, which is what the captured optimizer update then reads on every replay."""
        self.exchange_static(grads, guard)

    def exchange_static(self, grads, guard=None):
        """static gradient tensors -> flat buffer (one multi-tensor copy; zeros where there is no gradient), ONE all-reduce of the
        whole buffer incl. the guard slot (the backward has finished: nothing to overlap with), ``.grad`` re-pointed at the flat
        views.  A parameter without a gradient on THIS rank contributes zeros, so a parameter that received a gradient on one rank
        only still gets the mean over all ranks everywhere."""
        self._set_guard(guard)
        have = [(self._view[id(p)], g) for p, g in zip(self.params, grads) if g is not None]
        none = [self._view[id(p)] for p, g in zip(self.params, grads) if g is None]
        if have:
            torch._foreach_copy_([v for v, _ in have], [g for _, g in have])
        if none:
            torch._foreach_zero_(none)
        dist.all_reduce(self.flat, op=self._op, group=self.group)
        if not self._avg:
            self.flat.div_(self.world)
        for p in self.params:
            p.grad = self._view[id(p)]

    @property
    def grad_bytes(self) -> int:
        return self._numel * self._esize


def shard_batch(batch, rank: int, world: int):
    """Contiguous split of every [B, ...] tensor / per-sample list of a batch dict (recursive)."""
    def cut(v, B):
        lo, hi = rank * B // world, (rank + 1) * B // world
        return v[lo:hi]
    B = batch["video"].shape[0]
    out = {}
    for k, v in batch.items():
        if isinstance(v, dict):
            out[k] = {kk: cut(vv, B) if hasattr(vv, "__len__") and len(vv) == B else vv for kk, vv in v.items()}
        elif hasattr(v, "__len__") and len(v) == B:
            out[k] = cut(v, B)
        else:
            out[k] = v
    return out
