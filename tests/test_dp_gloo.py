"""The N>1 path on CPU: 2 processes over gloo.  The flat-buffer gradient all-reduce must reproduce
the single-process gradient of the full batch (equal shards, mean losses), with and without
bucket overlap, and must broadcast rank 0's parameters."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

from shufflingvideosfortsg_amd.dp import FlatGradAllReduce, shard_batch


class Tiny(nn.Module):
    """LSTM + Linear + LayerNorm: the recurrent / dense glue of the grounding model (no HIP ops)."""

    def __init__(self):
        super().__init__()
        self.lstm = nn.LSTM(6, 5, 2, batch_first=True, bidirectional=True)
        self.norm = nn.LayerNorm(10)
        self.head = nn.Linear(10, 1)
        self.unused = nn.Linear(3, 3)          # never receives a gradient

    def forward(self, x):
        return torch.softmax(self.head(self.norm(self.lstm(x)[0])).squeeze(2), dim=1)


def _loss(model, batch):
    p = model(batch["video"])
    idx = torch.as_tensor(batch["gt"]["framestps"])[:, 0:1]
    return -torch.log(p.gather(1, idx)).mean()


def _batch(B=8, T=7):
    g = torch.Generator().manual_seed(3)
    return {"video": torch.randn(B, T, 6, generator=g), "gt": {"framestps": [[i % T, T - 1] for i in range(B)]}}


def _worker(rank, world, port, overlap, bucket_mb, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)              # different init per rank: broadcast must fix it
    model = Tiny()
    static = overlap == "static"               # the graph-replay protocol: static gradient tensors, exchange_static()
    dp = FlatGradAllReduce(model, bucket_mb=bucket_mb, overlap=False if static else overlap)
    local = shard_batch(_batch(), rank, world)
    dp.zero_grad()
    _loss(model, local).backward()
    if static:
        grads = [p.grad for p in dp.params]    # what a captured backward leaves behind (None: no gradient)
        dp.adopt(grads)
        views = [p.grad for p in dp.params]
        for g in grads:                        # a "replay": the same tensors are rewritten, the exchange runs again
            if g is not None:
                g.mul_(1.0)
        dp.exchange_static(grads)
        assert all(a is b or a.data_ptr() == b.data_ptr() for a, b in zip(views, (p.grad for p in dp.params)))
    else:
        dp.finish()
    if rank == 0:
        # numpy (pickled by value): tensors would travel as shared-memory fds that die with the worker
        out.put({k: p.grad.numpy().copy() for k, p in model.named_parameters()})
        out.put({k: v.numpy().copy() for k, v in model.state_dict().items()})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("overlap,bucket_mb", [(True, 0.0002), (False, 32.0), ("static", 32.0)])
def test_flat_allreduce_matches_single_process(overlap, bucket_mb):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, overlap, bucket_mb, q)) for r in range(2)]
    for p in procs:
        p.start()
    grads, state = ({k: torch.from_numpy(v) for k, v in d.items()} for d in (q.get(), q.get()))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    torch.manual_seed(100)                     # rank 0's init
    ref = Tiny()
    for k, v in ref.state_dict().items():
        torch.testing.assert_close(state[k], v, atol=0, rtol=0)
    _loss(ref, _batch()).backward()
    for k, p in ref.named_parameters():
        want = p.grad if p.grad is not None else torch.zeros_like(p)
        torch.testing.assert_close(grads[k], want, atol=1e-6, rtol=1e-5, msg=lambda m, k=k: f"{k}: {m}")


def test_shard_batch():
    b = _batch(8)
    parts = [shard_batch(b, r, 4) for r in range(4)]
    assert sum(p["video"].shape[0] for p in parts) == 8
    assert torch.equal(torch.cat([p["video"] for p in parts]), b["video"])
    assert sum((p["gt"]["framestps"] for p in parts), []) == b["gt"]["framestps"]
