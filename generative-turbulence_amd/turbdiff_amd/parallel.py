"""Data-parallel training and trajectory-sharded sampling: one process per GPU,
``torch.distributed`` over RCCL/xGMI (backend "nccl" on ROCm; "gloo" in the CPU tests).

The reference has no distributed code (every config sets ``devices: 1``); this
module is new capability.  Design for xGMI (point-to-point links, ring collectives are per-
link bound): few large all-reduces (default 48 MB buckets) issued from autograd hooks as soon
as a bucket's last gradient is ready, so they overlap with the rest of backward.  The U-Net's
parameter mass sits in the middle of the network (center + down.3 + up.0 hold ~87 %), whose
gradients become ready when roughly half of backward -- all the high-resolution down path --
is still to run, which is what hides the collective.

Buckets follow the order in which gradients became ready during the first backward pass
(decode -> up -> center -> down -> encoders), not the registration order.
"""

from __future__ import annotations

import torch
import torch.distributed as dist


def init_from_env(backend: str | None = None):
    """Initialise the default process group from RANK / WORLD_SIZE / MASTER_* (torchrun).
    Returns (rank, world_size, local_rank).  No-op for single-process runs."""
    import os

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


class BucketedDataParallel:
    """Gradient averaging for a module whose backward runs custom autograd Functions.

        ddp = BucketedDataParallel(model)          # broadcasts parameters from rank 0
        loss.backward()                            # hooks launch async all-reduces
        ddp.finish()                               # wait; p.grad <- mean over ranks
        optimizer.step()
    """

    def __init__(self, module: torch.nn.Module, bucket_mb: float = 48.0, group=None, broadcast: bool = True):
        self.module = module
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.params = [p for p in module.parameters() if p.requires_grad]
        self.bucket_bytes = int(bucket_mb * 2**20)
        self._order: list[int] = []          # ready order observed in the first backward
        self._buckets: list[list[int]] | None = None
        self._bucket_of: dict[int, int] = {}
        self._pending: list[int] = []
        self._flat: dict[int, torch.Tensor] = {}
        self._work: list = []
        self._pads: dict = {}
        self._launched: set[int] = set()
        if self.world > 1 and broadcast:
            for p in self.params:
                dist.broadcast(p.data, src=0, group=group)
            for b in module.buffers():
                dist.broadcast(b.data, src=0, group=group)
        if self.world > 1:
            for i, p in enumerate(self.params):
                p.register_post_accumulate_grad_hook(self._make_hook(i))

    # -- bucket construction ------------------------------------------------------------
    def _build_buckets(self, order: list[int]):
        seen = set(order)
        order = order + [i for i in range(len(self.params)) if i not in seen]  # unused params last
        buckets, cur, cur_bytes = [], [], 0
        for i in order:
            n = self.params[i].numel() * self.params[i].element_size()
            if cur and cur_bytes + n > self.bucket_bytes:
                buckets.append(cur)
                cur, cur_bytes = [], 0
            cur.append(i)
            cur_bytes += n
        if cur:
            buckets.append(cur)
        self._buckets = buckets
        self._bucket_of = {i: b for b, idxs in enumerate(buckets) for i in idxs}
        self._pending = [len(b) for b in buckets]

    def bucket_layout(self):
        """[(n_params, bytes)] per bucket, in launch order (None before the first backward)."""
        if self._buckets is None:
            return None
        return [(len(b), sum(self.params[i].numel() * self.params[i].element_size() for i in b)) for b in self._buckets]

    # -- hooks ----------------------------------------------------------------------------
    def _make_hook(self, i: int):
        def hook(param):
            if self._buckets is None:
                self._order.append(i)  # first pass: only record the order
                return
            b = self._bucket_of[i]
            self._pending[b] -= 1
            if self._pending[b] == 0:
                self._launch(b)

        return hook

    def _pad(self, like: torch.Tensor, n: int) -> torch.Tensor:
        key = (like.device, like.dtype, n)
        z = self._pads.get(key)
        if z is None:
            z = self._pads[key] = torch.zeros(n, dtype=like.dtype, device=like.device)
        return z

    def _launch(self, b: int):
        idxs = self._buckets[b]
        grads = [self.params[i].grad if self.params[i].grad is not None else torch.zeros_like(self.params[i]) for i in idxs]
        # every slice starts on a 16-byte boundary (zero pads in between), so that the views handed back
        # as .grad keep the alignment the fused optimiser's 16-B accesses want
        parts = []
        for g in grads:
            parts.append(g.reshape(-1))
            r = (-g.numel()) % 4
            if r:
                parts.append(self._pad(g, r))
        flat = torch.cat(parts)
        flat.div_(self.world)
        self._flat[b] = flat
        self._work.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        self._launched.add(b)

    # -- per-step API ----------------------------------------------------------------------
    def finish(self):
        """Complete the step's gradient exchange.  Must be called after backward()."""
        if self.world == 1:
            return
        if self._buckets is None:
            # first step: gradients are all there; build buckets in the observed order and
            # reduce them now (no overlap on this step only)
            self._build_buckets(self._order)
        for b in range(len(self._buckets)):
            if b not in self._launched:
                self._launch(b)  # parameters that received no gradient this step
        for w in self._work:
            w.wait()
        for b, idxs in enumerate(self._buckets):
            flat, off = self._flat[b], 0
            for i in idxs:
                p = self.params[i]
                n = p.numel()
                p.grad = flat[off : off + n].view_as(p)
                off += n + (-n) % 4
        self._work.clear()
        self._flat.clear()
        self._launched.clear()
        self._pending = [len(b) for b in self._buckets]


def shard_trajectories(n_total: int, rank: int, world: int) -> range:
    """Contiguous block of trajectory ids owned by `rank` (sampling is embarrassingly
    parallel: no collective inside the loop)."""
    per, extra = divmod(n_total, world)
    start = rank * per + min(rank, extra)
    return range(start, start + per + (1 if rank < extra else 0))
