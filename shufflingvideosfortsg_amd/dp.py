"""Data parallelism for the grounding models: one process per GPU, full replica per rank, ONE flat
fp32 gradient buffer that is all-reduced (sum) over RCCL/xGMI in a few large buckets while the
backward is still running, then scaled by 1/world.  (The reference only has a single-process
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
    """Owns the gradients of ``module`` as views into one contiguous buffer and averages them
    across ranks.  Usage per step:  ``dp.zero_grad(); loss.backward(); dp.finish(); optim.step()``.

    bucket_mb: target bucket size.  xGMI is point-to-point (7 links/GPU), so few large ring
    reductions beat many small ones; 32 MiB buckets give ~6 buckets for the 186 MB d=1024 model.
    """

    def __init__(self, module: torch.nn.Module, process_group: Optional[dist.ProcessGroup] = None,
                 bucket_mb: float = 32.0, overlap: bool = True, broadcast: bool = True):
        self.module = module
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.params: List[torch.nn.Parameter] = [p for p in module.parameters() if p.requires_grad]
        if not self.params:
            raise ValueError("module has no trainable parameters")
        dev, dt = self.params[0].device, self.params[0].dtype
        # backward produces gradients roughly in reverse registration order: lay the buffer out that
        # way so that each bucket is a contiguous slice that completes early
        order = list(reversed(self.params))
        total = sum(p.numel() for p in order)
        self.flat = torch.zeros(total, device=dev, dtype=dt)
        self.buckets = []          # (start, end, n_params)
        cap = max(1, int(bucket_mb * (1 << 20) / self.flat.element_size()))
        off = start = count = 0
        self._bucket_of = {}
        for p in order:
            n = p.numel()
            p.grad = self.flat[off:off + n].view_as(p)
            self._bucket_of[id(p)] = len(self.buckets)
            off += n; count += 1
            if off - start >= cap:
                self.buckets.append((start, off, count)); start, count = off, 0
        if count:
            self.buckets.append((start, off, count))
        self._ready = [0] * len(self.buckets)
        self._handles = []
        self._force = dist.is_initialized() and __import__("os").environ.get("TSG_FORCE_DIST") == "1"
        self.overlap = overlap and (self.world > 1 or self._force)
        if self.overlap:
            for p in self.params:
                p.register_post_accumulate_grad_hook(self._hook)
        if broadcast and self.world > 1:
            for t in list(module.parameters()) + list(module.buffers()):
                dist.broadcast(t.data, src=0, group=self.group)

    # -- per-step protocol ----------------------------------------------------------------
    def zero_grad(self):
        self.flat.zero_()
        for p in self.params:                     # an optimizer may have detached the views
            if p.grad is None or p.grad.data_ptr() < self.flat.data_ptr() or \
                    p.grad.data_ptr() >= self.flat.data_ptr() + self.flat.numel() * self.flat.element_size():
                raise RuntimeError("a parameter's .grad no longer aliases the flat buffer "
                                   "(use dp.zero_grad(), not optimizer.zero_grad(set_to_none=True))")
        self._ready = [0] * len(self.buckets)
        self._handles = []

    def _launch(self, b):
        s, e, _ = self.buckets[b]
        self._handles.append(dist.all_reduce(self.flat[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def _hook(self, p):
        b = self._bucket_of[id(p)]
        self._ready[b] += 1
        if self._ready[b] == self.buckets[b][2]:
            self._launch(b)

    def finish(self):
        """Complete the gradient exchange: afterwards every rank holds the mean gradient."""
        if self.world == 1 and not self._force:
            return
        if self.overlap:
            for b, (s, e, n) in enumerate(self.buckets):   # parameters that received no gradient
                if self._ready[b] != n:
                    self._launch(b)
        else:
            self._handles = [dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)]
        for h in self._handles:
            h.wait()
        self.flat.div_(self.world)

    @property
    def grad_bytes(self) -> int:
        return self.flat.numel() * self.flat.element_size()


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
