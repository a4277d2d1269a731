"""``FlatClipAdam``: gradient clipping + Adam on one flat fp32 bucket, three HIP launches per step.

Reference training step (libs/utils_ft.py:676-681)::

    loss.backward(); nn.utils.clip_grad_norm_(model.parameters(), grad_clip); optimizer.step()

with ``torch.optim.Adam``.  Here the parameters live in ONE contiguous fp32 buffer (every ``p.data`` is a view of
it, so ``state_dict`` / ``load_state_dict`` / the modules are unaffected), the gradients are gathered into a second
flat buffer -- which is also the bucket the data-parallel all-reduce works on, so there is no copy back -- and
``gt_grad_sqnorm`` + ``gt_adam_clip_step`` (csrc/gt_optim.hip) do norm, clip coefficient, moments and update.  The step
count and the learning rate are device scalars: a captured HIP graph of the step stays valid while a scheduler changes
the rate.  It is a ``torch.optim.Optimizer``, so ``OneCycleLR`` and ``run_train`` take it unchanged.
"""
from __future__ import annotations

from typing import Iterable, Optional

import torch
import torch.distributed as dist

from . import _hip as H


class FlatClipAdam(torch.optim.Optimizer):
    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 0.0, max_norm: Optional[float] = None, group=None, model: Optional[torch.nn.Module] = None):
        """model (optional): modules that concatenate some of their parameters on every forward (SimpleAttention: the packed
        QKV weight / bias and the per-head LayerNorm parameters) publish those groups through ``_pack_groups()``; the groups
        are laid out back to back in the bucket, so the module gets its packed tensor as a VIEW of the bucket instead of four
        ``torch.cat`` / ``torch.stack`` launches per layer and step (ops.packed_params)."""
        params = [p for p in params if p.requires_grad]
        if not params:
            raise ValueError("FlatClipAdam: no trainable parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.params = params
        dev = params[0].device
        # the flat buckets (and the collective on them) are device-agnostic -- the multi-rank CPU tests run them over
        # gloo; the update itself is HIP only: apply() refuses CPU tensors (no CPU fallback for the optimizer step)
        if any(p.dtype != torch.float32 for p in params):
            raise TypeError("FlatClipAdam: fp32 parameters")
        if any(p.device != dev for p in params):
            raise ValueError("FlatClipAdam: all parameters must live on one device")
        self.max_norm = float(max_norm) if max_norm else 0.0
        self.group = group
        self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        # every tensor starts on a 16-byte boundary of the bucket (the kernels use 16-byte accesses); the members of a pack
        # group follow each other without padding (their sizes must then be multiples of four -- else the group is ignored)
        order, placed = [], set()
        index = {id(p): i for i, p in enumerate(params)}
        if model is not None:
            for m in model.modules():
                for grp in (m._pack_groups() if hasattr(m, "_pack_groups") else []):
                    ids = [index.get(id(p)) for p in grp]
                    if any(i is None or i in placed for i in ids) or any(p.numel() % 4 for p in grp):
                        continue
                    order.extend(ids)
                    placed.update(ids)
        order.extend(i for i in range(len(params)) if i not in placed)
        self.offsets, off = [0] * len(params), 0
        for i in order:
            self.offsets[i] = off
            off += (params[i].numel() + 3) // 4 * 4
        self.numel = off
        f32 = dict(dtype=torch.float32, device=dev)
        self.flat_param = torch.zeros(off, **f32)
        self.flat_grad = torch.zeros(off, **f32)
        self.exp_avg = torch.zeros(off, **f32)
        self.exp_avg_sq = torch.zeros(off, **f32)
        self.grad_views = []
        with torch.no_grad():
            for p, o in zip(params, self.offsets):
                v = self.flat_param[o:o + p.numel()].view_as(p)
                v.copy_(p.data)
                p.data = v                                    # the module now reads / the kernel updates the bucket
                self.grad_views.append(self.flat_grad[o:o + p.numel()].view_as(p))
        self.step_count = torch.zeros(1, dtype=torch.int64, device=dev)      # device-resident (uint64 in the ABI)
        self.sched_dev = torch.tensor([float(lr), float(betas[0])], **f32)    # what schedulers move: lr, beta1
        self.sqnorm = torch.zeros(1, **f32)
        self._sched_host = (float(lr), float(betas[0]))
        self._built = True
        self._bind_state()

    # ---- optimizer state: the moments are exposed per parameter (views of the flat buffers) in torch.optim.Adam's
    # layout, so state_dict() / load_state_dict() -- what run_train pickles as `optimizer_state` -- round-trip them
    def _bind_state(self):
        for p, o in zip(self.params, self.offsets):
            n = p.numel()
            self.state[p] = {"step": self.step_count, "exp_avg": self.exp_avg[o:o + n].view_as(p),
                             "exp_avg_sq": self.exp_avg_sq[o:o + n].view_as(p)}

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)              # fills self.state[p] with (copies of) the saved tensors
        steps = []
        with torch.no_grad():
            for p, o in zip(self.params, self.offsets):
                st, n = self.state.get(p), p.numel()
                if not st:
                    continue
                self.exp_avg[o:o + n].copy_(st["exp_avg"].reshape(-1))
                self.exp_avg_sq[o:o + n].copy_(st["exp_avg_sq"].reshape(-1))
                steps.append(int(torch.as_tensor(st["step"]).reshape(-1)[0].item()))
            self.step_count.fill_(max(steps) if steps else 0)
        self._bind_state()
        self._sched_host = None                           # param_groups may carry another lr / beta1
        self.sync_schedule()

    def add_param_group(self, param_group):
        if getattr(self, "_built", False) or len(self.param_groups) >= 1:
            raise NotImplementedError("FlatClipAdam keeps ONE flat bucket with one set of hyper-parameters: "
                                      "per-group options / add_param_group are not supported")
        super().add_param_group(param_group)

    def _check_views(self):
        """Every p.data must still be the view of the flat parameter bucket made at construction: model.to() / .float()
        / a manual `p.data = ...` rebinds it, and the optimizer would then update a bucket nobody reads."""
        base, end = self.flat_param.data_ptr(), self.flat_param.data_ptr() + 4 * self.numel
        for p, o in zip(self.params, self.offsets):
            if p.data_ptr() != base + 4 * o or not (base <= p.data_ptr() < end):
                raise RuntimeError("FlatClipAdam: a parameter no longer lives in the optimizer's flat bucket (the model was "
                                   "moved / cast / re-bound after the optimizer was built); rebuild the optimizer")

    # ---- pieces (bench.py / a DDP loop call them separately to put the all-reduce in between) -----------------
    def gather_grads(self):
        """flat_grad <- the parameters' .grad: one multi-tensor copy.  A parameter without a gradient counts as a zero
        gradient (its moments decay and weight decay applies), where torch.optim.Adam would skip it: every parameter of
        the models here takes a gradient in every step."""
        srcs, dsts = [], []
        for p, v in zip(self.params, self.grad_views):
            if p.grad is None:
                v.zero_()
            else:
                srcs.append(p.grad)
                dsts.append(v)
        if srcs:
            torch._foreach_copy_(dsts, srcs)

    def all_reduce(self):
        """Sum the flat gradient over the ranks (the 1/world average is folded into the update)."""
        if self.world > 1:
            dist.all_reduce(self.flat_grad, op=dist.ReduceOp.SUM, group=self.group)

    def sync_schedule(self):
        """Copy the scheduler-controlled hyper-parameters (lr, and beta1, which OneCycleLR cycles as Adam's momentum)
        from param_groups[0] to their device scalars.  Call it outside a captured graph; the graph reads the scalars."""
        g = self.param_groups[0]
        cur = (float(g["lr"]), float(g["betas"][0]))
        if len(self.param_groups) != 1:
            raise NotImplementedError("FlatClipAdam: one parameter group")
        if cur != self._sched_host:
            self.sched_dev.copy_(torch.tensor(cur, dtype=torch.float32), non_blocking=True)
            self._sched_host = cur

    def apply(self):
        """norm -> clip -> Adam on the flat buffers (capturable: no host read-back)."""
        H.need_f32_cuda(self.flat_param, self.flat_grad)
        H.weight_packs.invalidate()              # the weights change below: packs made ahead of them are stale
        g = self.param_groups[0]
        L = H.lib()
        st = H.stream_ptr()
        gscale = 1.0 / self.world
        H.check(L.gt_seed_advance(self.step_count.data_ptr(), 1, st), "gt_seed_advance")
        if self.max_norm > 0:
            ws = H.workspace(self.flat_grad.device, L.gt_grad_sqnorm_ws_bytes())
            H.check(L.gt_grad_sqnorm(self.flat_grad.data_ptr(), self.numel, gscale, self.sqnorm.data_ptr(),
                                     ws.data_ptr(), ws.numel(), st), "gt_grad_sqnorm")
        H.check(L.gt_adam_clip_step(self.flat_param.data_ptr(), self.flat_grad.data_ptr(), self.exp_avg.data_ptr(),
                                    self.exp_avg_sq.data_ptr(), self.numel,
                                    self.sqnorm.data_ptr() if self.max_norm > 0 else None, gscale, self.max_norm,
                                    self.sched_dev.data_ptr(), g["betas"][0], g["betas"][1], g["eps"],
                                    g["weight_decay"], self.step_count.data_ptr(),
                                    self.sched_dev.data_ptr() + 4, st), "gt_adam_clip_step")

    # ---- torch.optim.Optimizer interface ----------------------------------------------------------------------
    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self._check_views()
        self.sync_schedule()                                  # schedulers write param_groups[0]
        self.gather_grads()
        self.all_reduce()
        self.apply()
        return loss

    def grad_norm(self) -> float:
        """Global norm of the (averaged) gradient seen by the last step (host read-back; diagnostics only)."""
        return float(self.sqnorm.sqrt().item())
