"""Fused tail of the training step: global-norm gradient clipping + RAdam in three kernel launches.

The reference trains with ``torch.optim.RAdam(lr)`` (models/diffusion.py:210-218) under Lightning's
``gradient_clip_val: 0.1`` (config/train.yaml:30-31), i.e. ``clip_grad_norm_`` followed by
``optimizer.step()``.  On 149 separately allocated parameter tensors the stock foreach path costs
~40 multi-tensor launches and ~70 small fills per step (1.6 ms of a 33 ms step on MI355X);
``ClipRAdam`` does the same arithmetic (torch.optim.radam._single_tensor_radam, same scalar schedule,
fp32 state) through ``tdx_grad_norm`` / ``tdx_radam_step`` (csrc/tdx_optim.hip).

It is a ``torch.optim.Optimizer``: ``param_groups[i]["lr"]`` is honoured every step (LR schedulers
work), and the per-parameter state uses torch's RAdam keys (``step``, ``exp_avg``, ``exp_avg_sq``) with
per-parameter step counts (parameters at different counts are updated in one launch per count), so
optimizer state_dicts move between the two implementations.

Loss scaling (fp16 training, ``loss_scale=``): the backward pass runs on ``scale_loss(loss) = S * loss`` (S a power of two) so
that fp16 activation gradients stay representable; the stored fp32 parameter gradients are then S times the true ones.
``tdx_grad_norm_scaled`` reports the true norm, folds 1 / S into the clip factor and raises a device flag when the norm is
not finite, on which ``tdx_radam_step_scaled`` leaves parameters and moments untouched.  The host never waits for that flag:
it reads the flag of step k - 2 at the END of step k (an asynchronous copy into pinned memory behind an event that has long fired),
halves S and takes the skipped step out of the step counters then; after ``scale_growth_interval`` clean steps S doubles.
The lag is a constant two steps, so data-parallel ranks (which see identical all-reduced gradients) change S in lockstep.
Its price: the one or two steps that follow an overflow run RAdam's bias corrections with a step count that still includes
the skipped step(s) (``sync_flags=True`` waits for every step's own flag instead, as ``torch.cuda.amp.GradScaler`` does).
"""

from __future__ import annotations

import numpy as np
import torch

from . import _lib as L


class ClipRAdam(torch.optim.Optimizer):
    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, max_norm: float | None = None,
                 write_clipped_grads: bool = False, loss_scale: float | None = None, scale_growth_interval: int = 2000,
                 min_loss_scale: float = 1.0, max_loss_scale: float = 2.0**24, sync_flags: bool = False):
        if not 0.0 <= lr:
            raise ValueError(f"Invalid learning rate: {lr}")
        if not (0.0 <= betas[0] < 1.0 and 0.0 <= betas[1] < 1.0):
            raise ValueError(f"Invalid betas: {betas}")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=0.0))
        self.max_norm = max_norm
        self.write_clipped_grads = write_clipped_grads
        self._plans = {}
        # loss scaling (None: gradients are taken as they are)
        if loss_scale is not None and (loss_scale <= 0 or 2.0 ** round(np.log2(loss_scale)) != loss_scale):
            raise ValueError(f"loss_scale must be a positive power of two, got {loss_scale}")
        self.loss_scale = None if loss_scale is None else float(loss_scale)
        self.scale_growth_interval = int(scale_growth_interval)
        self.min_loss_scale, self.max_loss_scale = float(min_loss_scale), float(max_loss_scale)
        # True: every step waits for its own flag (torch.cuda.amp.GradScaler's behaviour: one host sync per step, and the
        # step counters of the steps right after an overflow are exact instead of one or two too high until the flag lands)
        self.sync_flags = bool(sync_flags)
        self.skipped_steps = 0      # steps whose gradients were not finite (known with the two-step lag)
        self._clean_steps = 0       # since the last change of the scale
        self._flags = []            # [(step index, pinned (4,) float tensor, event, plan index)] not yet looked at
        self._step_index = 0

    def scale_loss(self, loss: torch.Tensor) -> torch.Tensor:
        """``loss * S`` for the backward pass (the identity without loss scaling): ``opt.scale_loss(loss).backward()``."""
        return loss if self.loss_scale is None else loss * self.loss_scale

    def _settle_flags(self, upto: int, wait: bool) -> None:
        """Look at the non-finite flags of the steps < `upto` (in order): a skipped step leaves the step counters and
        halves the scale; `scale_growth_interval` clean steps in a row double it."""
        while self._flags and self._flags[0][0] < upto:
            idx, host, ev, gi = self._flags[0]
            if not wait and not ev.query():
                break
            ev.synchronize()
            self._flags.pop(0)
            if float(host[2]) != 0.0:
                self.skipped_steps += 1
                self._clean_steps = 0
                self.loss_scale = max(self.loss_scale * 0.5, self.min_loss_scale)
                plan = self._plans.get(gi)
                if plan is not None and plan.get("fixed") is not None:
                    plan["fixed"][2].sub_(plan["live_mask"][idx % 4])  # that step never happened for the moments
            else:
                self._clean_steps += 1
                if self.scale_growth_interval > 0 and self._clean_steps >= self.scale_growth_interval:
                    self._clean_steps = 0
                    self.loss_scale = min(self.loss_scale * 2.0, self.max_loss_scale)

    def settle(self) -> None:
        """Wait for every outstanding non-finite flag (end of training, tests, checkpoints): afterwards `loss_scale`,
        `skipped_steps` and the per-parameter step counts reflect all steps taken."""
        if self.loss_scale is not None:
            self._settle_flags(self._step_index + 1, wait=True)

    # one plan per parameter group: chunk tables (static) + pointer table (refreshed every step)
    def _plan(self, gi, group):
        plan = self._plans.get(gi)
        params = [p for p in group["params"]]
        if plan is not None and plan["n"] == len(params):
            return plan
        if not params:
            return None
        dev = params[0].device
        for p in params:
            if p.dtype != torch.float32 or not p.is_contiguous() or p.device != dev or not p.is_cuda:
                raise TypeError("ClipRAdam needs contiguous float32 parameters on one GPU")
            st = self.state[p]
            if len(st) == 0:
                st["step"] = torch.tensor(0.0, dtype=torch.float32)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        chunk = L.query("tdx_opt_chunk_elems")
        ct, co = [], []
        for i, p in enumerate(params):
            for o in range(0, p.numel(), chunk):
                ct.append(i)
                co.append(o)
        # pinned staging for the pointer table, rotated so that a buffer is never rewritten while an
        # earlier step's asynchronous copy may still be reading it
        hosts = [torch.zeros((len(params), 5), dtype=torch.int64).pin_memory() for _ in range(3)]
        plan = {
            "n": len(params), "params": params, "dev": dev, "nchunks": len(ct),
            "chunk_tensor": torch.tensor(ct, dtype=torch.int32, device=dev),
            "chunk_off": torch.tensor(co, dtype=torch.int64, device=dev),
            "hosts": hosts, "events": [None] * 3, "turn": 0,
            "table": torch.empty((len(params), 5), dtype=torch.int64, device=dev),
            "partial": torch.empty(len(ct), dtype=torch.float32, device=dev),
            "norm": torch.zeros(4, dtype=torch.float32, device=dev),
            # loss scaling: pinned landing slots of the last four steps' (norm, factor, non-finite flag) and which
            # parameters those steps counted (to take a skipped step back out of the counters)
            "flag_hosts": [torch.zeros(4, dtype=torch.float32).pin_memory() for _ in range(4)],
            "live_mask": [torch.zeros(len(params), dtype=torch.float32) for _ in range(4)],
        }
        self._plans[gi] = plan
        return plan

    @staticmethod
    def _chunks_of(plan, members):
        """Chunk tables restricted to the parameters `members` (cached per member set)."""
        cache = plan.setdefault("subsets", {})
        hit = cache.get(members)
        if hit is None:
            keep = torch.isin(plan["chunk_tensor"], torch.tensor(members, dtype=torch.int32, device=plan["dev"]))
            hit = cache[members] = (plan["chunk_tensor"][keep].contiguous(), plan["chunk_off"][keep].contiguous(), int(keep.sum()))
        return hit

    @torch.no_grad()
    def step(self, closure=None):
        """clip (if ``max_norm``) + RAdam.  Returns the loss of ``closure`` if given; the total gradient
        norm of the last step stays on the device in ``self.last_grad_norm`` (no host sync)."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if len(self.param_groups) > 1 and (self.max_norm or self.loss_scale is not None):
            raise NotImplementedError("global-norm clipping / loss scaling over several parameter groups")
        scaled = self.loss_scale is not None
        if scaled:
            self._step_index += 1
        for gi, group in enumerate(self.param_groups):
            plan = self._plan(gi, group)
            if plan is None:
                continue
            k = plan["turn"] = (plan["turn"] + 1) % 3
            if plan["events"][k] is not None:
                plan["events"][k].synchronize()
            params = plan["params"]
            grads = [p.grad for p in params]
            if any(g is not None and (g.dtype != torch.float32 or not g.is_contiguous()) for g in grads):
                for p, g in zip(params, grads):
                    if g is not None and (g.dtype != torch.float32 or not g.is_contiguous()):
                        p.grad = g.float().contiguous()
                grads = [p.grad for p in params]
            # the host's share of a step matters once forward + backward replay from a captured graph: the static columns
            # (parameter / moment addresses, sizes) are written once per plan, the gradient column as one array, and the
            # per-parameter step counters (0-dim views into ONE tensor, so state_dict keeps torch's layout) with one add
            fixed = plan.get("fixed")
            state = [self.state[p] for p in params]
            if (fixed is None or any(st["exp_avg"] is not a or st["step"] is not v for st, a, v in zip(state, fixed[1], fixed[3]))
                    or [p.data_ptr() for p in params] != fixed[4]):
                # first step, or somebody replaced state tensors (load_state_dict) or a parameter's storage
                # (model.to(...), p.data = ..., load_state_dict(assign=True)): adopt them
                cols = np.array([[p.data_ptr(), 0, st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel()]
                                 for p, st in zip(params, state)], dtype=np.int64)
                steps = torch.tensor([float(st["step"]) for st in state], dtype=torch.float32)
                views = [steps[i] for i in range(len(params))]
                for st, v in zip(state, views):
                    st["step"] = v
                fixed = plan["fixed"] = (cols, [st["exp_avg"] for st in state], steps, views, [p.data_ptr() for p in params])
            tab = plan["hosts"][k].numpy()
            tab[:] = fixed[0]
            tab[:, 1] = [0 if g is None else g.data_ptr() for g in grads]
            steps = fixed[2]
            live_all = all(g is not None for g in grads)
            if live_all:
                steps += 1
            else:
                for i, g in enumerate(grads):
                    if g is not None:
                        steps[i] += 1
            if scaled:
                mask = plan["live_mask"][self._step_index % 4]
                if live_all:
                    mask.fill_(1.0)
                else:
                    mask.copy_(torch.tensor([0.0 if g is None else 1.0 for g in grads]))
            plan["table"].copy_(plan["hosts"][k], non_blocking=True)
            plan["events"][k] = torch.cuda.Event()
            plan["events"][k].record()
            live = [i for i, g in enumerate(grads) if g is not None]
            if not live:
                continue
            stream = L.stream()
            clip = None
            if scaled:
                L.call("tdx_grad_norm_scaled", L.ptr(plan["table"]), L.ptr(plan["chunk_tensor"]), L.ptr(plan["chunk_off"]),
                       plan["nchunks"], float(self.max_norm or 0.0), 1.0 / self.loss_scale, L.ptr(plan["partial"]),
                       L.ptr(plan["norm"]), stream)
                clip = plan["norm"]
                self.last_grad_norm = plan["norm"][0]
                host = plan["flag_hosts"][self._step_index % 4]
                host.copy_(plan["norm"], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
                self._flags.append((self._step_index, host, ev, gi))
            elif self.max_norm:
                L.call("tdx_grad_norm", L.ptr(plan["table"]), L.ptr(plan["chunk_tensor"]), L.ptr(plan["chunk_off"]),
                       plan["nchunks"], float(self.max_norm), L.ptr(plan["partial"]), L.ptr(plan["norm"]), stream)
                clip = plan["norm"]
                self.last_grad_norm = plan["norm"][0]
            b1, b2 = group["betas"]
            # RAdam's bias corrections depend on the step count, which torch keeps per parameter: parameters that
            # received gradients on different numbers of steps (conditional branches, a state_dict loaded from
            # torch.optim.RAdam) are updated in one launch per distinct count, over that count's chunks only
            by_step = {}
            counts = steps.tolist()
            if live_all and counts.count(counts[0]) == len(counts):
                by_step[int(counts[0])] = live
            else:
                for i in live:
                    by_step.setdefault(int(counts[i]), []).append(i)
            for step_count, members in by_step.items():
                if len(by_step) == 1:
                    ct, co, n = plan["chunk_tensor"], plan["chunk_off"], plan["nchunks"]
                else:
                    ct, co, n = self._chunks_of(plan, tuple(members))
                L.call("tdx_radam_step_scaled" if scaled else "tdx_radam_step", L.ptr(plan["table"]), L.ptr(ct), L.ptr(co), n,
                       L.ptr(clip), step_count, float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                       int(self.write_clipped_grads), stream)
            # the kernels wrote through raw pointers: tell autograd (and the packed-weight caches keyed
            # on Tensor._version, ops._packed_conv3) that the parameters changed
            torch.autograd.graph.increment_version([p for p in plan["params"] if p.grad is not None])
        if scaled:
            # only now may the scale change: this step's backward ran on the scale its kernels above divided by.  The flags of
            # steps <= k - 2 are long on the host (a constant lag: data-parallel ranks change the scale in lockstep)
            self._settle_flags(self._step_index + 1 if self.sync_flags else self._step_index - 1, wait=True)
        return loss
