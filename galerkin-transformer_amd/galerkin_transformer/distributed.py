"""Batch-sharded data parallelism for one node of MI355X GPUs: one process per GPU,
``torch.distributed`` with backend "nccl" (= RCCL over xGMI on ROCm); "gloo" on CPU for tests.

The reference is single-process.  Every sample of the hot path is independent
in the forward pass (all reductions -- per-head LayerNorm, K^T V over tokens, FFTs -- are within a
sample, no BatchNorm), so the only exchange step is ONE sum-all-reduce of the flat gradient
(2 220 829 fp32 = 8.9 MB for the Darcy-141 model) per step.  The payload is latency- rather than
link-bound, so it goes out as a single flat bucket: one collective, all seven xGMI links busy.
``clip_grad_norm_`` needs the *global* norm; after the all-reduce every rank holds identical
gradients, so the local norm is the global norm and no second collective is needed.
"""
from __future__ import annotations

import os
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


def init_distributed(backend: Optional[str] = None, device: Optional[torch.device] = None):
    """Join the process group described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT
    (torch.distributed.run exports them).  Returns (rank, local_rank, world_size)."""
    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local)
            kw["device_id"] = device or torch.device("cuda", local)
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, local, world


def broadcast_parameters(module: torch.nn.Module, src: int = 0):
    """Make every rank start from rank ``src``'s parameters and buffers (one flat broadcast)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    tensors = [p.data for p in module.parameters()] + [b.data for b in module.buffers()
                                                       if b.is_floating_point()]
    if not tensors:
        return
    flat = torch.cat([t.reshape(-1) for t in tensors])
    dist.broadcast(flat, src)
    off = 0
    for t in tensors:
        n = t.numel()
        t.copy_(flat[off:off + n].view_as(t))
        off += n


class FlatGradAllReducer:
    """Average the gradients of ``params`` across ranks through one flat fp32 bucket."""

    def __init__(self, params: Iterable[torch.nn.Parameter], group=None):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.group = group
        self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        self.sizes = [p.numel() for p in self.params]
        self.flat: Optional[torch.Tensor] = None
        self.views: List[torch.Tensor] = []

    def _ensure(self, like: torch.Tensor):
        if self.flat is None or self.flat.device != like.device:
            self.flat = torch.empty(sum(self.sizes), dtype=torch.float32, device=like.device)
            self.views = list(self.flat.split(self.sizes))

    @property
    def nbytes(self) -> int:
        return 4 * sum(self.sizes)

    def reduce(self):
        """grad <- mean over ranks of grad, in place.  No-op for a single process."""
        if self.world == 1:
            return
        grads = []
        for p in self.params:
            if p.grad is None:
                p.grad = torch.zeros_like(p)
            grads.append(p.grad)
        self._ensure(grads[0])
        torch._foreach_copy_(self.views, [g.reshape(-1) for g in grads])
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
        self.flat.mul_(1.0 / self.world)
        torch._foreach_copy_(grads, [v.view_as(g) for v, g in zip(self.views, grads)])


def shard_range(n_items: int, rank: int, world: int, drop_last: bool = True):
    """Contiguous, equally sized shard [lo, hi) of ``n_items`` for this rank (DistributedSampler with
    drop_last, as ex2_darcy.py:31-32 drops the ragged tail)."""
    per = n_items // world if drop_last else -(-n_items // world)
    lo = rank * per
    return lo, min(n_items, lo + per)


def rank_seed(base_seed: int, rank: int) -> int:
    """Per-rank dropout stream: identical parameter init comes from the shared torch seed, the
    dropout masks must differ between ranks."""
    return (base_seed + 7919 * rank) & 0x7FFFFFFFFFFFFFFF
