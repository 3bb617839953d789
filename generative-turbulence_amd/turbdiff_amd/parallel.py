"""Data-parallel training and trajectory-sharded sampling: one process per GPU,
``torch.distributed`` over RCCL/xGMI (backend "nccl" on ROCm; "gloo" in the CPU tests).

The reference has no distributed code (every config sets ``devices: 1``); this
module is new capability.  Design for xGMI (point-to-point links, ring collectives are per-
link bound): few large all-reduces (default 48 MB buckets) issued from autograd hooks as soon
as a bucket's last gradient is ready, so they overlap with the rest of backward.  The U-Net's
parameter mass sits in the middle of the network (center + down.3 + up.0 hold ~87 %), whose
gradients become ready when roughly half of backward -- all the high-resolution down path --
is still to run, which is what hides the collective.

Buckets are PERSISTENT flat buffers, one set per (dtype, device), laid out in the order in which
gradients become ready in backward (decode -> up -> center -> down -> encoders).  That order is
taken from the module (``grad_ready_order()``, see ``models.ddpm.DenoisingModel``) when it offers
one -- so the very first step already overlaps -- and otherwise observed during the first backward.
A parameter's hook scales its fresh gradient by 1 / world straight into its slice of the bucket (one
pass, no ``torch.cat``, no allocation) and re-points ``p.grad`` at that slice; the all-reduce runs in
place, so after ``finish()`` the optimiser reads the averaged gradients through the same views.

Contract: exactly ONE backward per ``finish()`` (no gradient accumulation across backwards), equal
per-rank batch sizes (ranks are averaged with equal weight; ``data.ofles.OpenFOAMSampler`` pads
the shards to equal length).
"""

from __future__ import annotations

import time

import torch
import torch.distributed as dist


def init_from_env(backend: str | None = None, force: bool = False):
    """Initialise the default process group from RANK / WORLD_SIZE / MASTER_* (torchrun).
    Returns (rank, world_size, local_rank).  A single-process run creates no group unless
    ``force`` (then a world-size-1 group: the communicator path runs without a second GPU)."""
    import os

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or force) and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


CUS_PER_GPU = 256  # MI355X: 8 XCDs x 32 CUs


def check_cu_budget(environ=None, cus_per_gpu: int = CUS_PER_GPU) -> tuple[int, int]:
    """The persistent conv kernels (ring forward / data gradient, producer-consumer weight gradient) launch one
    workgroup per CU on TDX_PERSISTENT_CUS CUs (default: all 256) and each needs a whole CU's LDS; RCCL's all-reduce
    kernels run one workgroup (= one CU) per channel.  When the two together ask for more CUs than the chip has, a
    persistent launch that finds CUs taken runs in two rounds (~2x its time) -- or the collective waits for a whole
    conv launch, and the overlap with backward is gone.  Raises unless TDX_PERSISTENT_CUS + NCCL_MAX_NCHANNELS <= 256
    (both must be SET for a multi-GPU run: RCCL's default channel count is its own choice).  Returns the two numbers."""
    import os

    env = os.environ if environ is None else environ
    cus = int(env.get("TDX_PERSISTENT_CUS", str(cus_per_gpu)))
    ch = env.get("NCCL_MAX_NCHANNELS")
    if ch is None:
        raise RuntimeError(
            "data-parallel training on GPUs: set NCCL_MAX_NCHANNELS (and TDX_PERSISTENT_CUS) so that the persistent conv "
            f"kernels and RCCL's channels fit the {cus_per_gpu} CUs together, e.g. TDX_PERSISTENT_CUS={cus_per_gpu - 32} "
            "NCCL_MAX_NCHANNELS=32 (what bench.py --gpus N sets)")
    if cus + int(ch) > cus_per_gpu:
        raise RuntimeError(f"TDX_PERSISTENT_CUS ({cus}) + NCCL_MAX_NCHANNELS ({ch}) > {cus_per_gpu} CUs: the persistent conv "
                           "kernels and the all-reduce would queue behind each other instead of overlapping")
    return cus, int(ch)


def _static_ready_order(module: torch.nn.Module, params: list) -> list[int] | None:
    """Indices into `params` in the order their gradients become ready, from the first submodule that
    offers ``grad_ready_order()`` (an iterable of parameters); parameters it does not list go last."""
    for m in module.modules():
        fn = getattr(m, "grad_ready_order", None)
        if callable(fn):
            index = {id(p): i for i, p in enumerate(params)}
            order, seen = [], set()
            for p in fn():
                i = index.get(id(p))
                if i is not None and i not in seen:
                    seen.add(i)
                    order.append(i)
            return order + [i for i in range(len(params)) if i not in seen]
    return None


class BucketedDataParallel:
    """Gradient averaging for a module whose backward runs custom autograd Functions.

        ddp = BucketedDataParallel(model)          # broadcasts parameters from rank 0
        loss.backward()                            # hooks launch async all-reduces
        ddp.finish()                               # wait; p.grad <- mean over ranks
        optimizer.step()

    compress="bf16": gradients travel as bfloat16 (half the bytes per xGMI link) and are widened
    back into the fp32 buckets the optimiser reads; off by default (changes the arithmetic).
    force=True keeps the hooks and collectives active at world size 1 (tests of the RCCL path).
    """

    def __init__(self, module: torch.nn.Module, bucket_mb: float = 48.0, group=None, broadcast: bool = True,
                 compress: str | None = None, force: bool = False):
        self.module = module
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.active = (self.world > 1 or force) and dist.is_initialized()
        self.enabled = True  # False: hooks do nothing (bench.py's no-communication reference leg)
        self.params = [p for p in module.parameters() if p.requires_grad]
        self.bucket_bytes = int(bucket_mb * 2**20)
        assert compress in (None, "bf16")
        self.compress = compress
        self._order: list[int] = []          # ready order observed in a backward (when no static order exists)
        self._buckets: list[list[int]] | None = None
        self._bucket_of: dict[int, int] = {}
        self._slot: dict[int, tuple[int, int]] = {}   # param index -> (bucket, offset)
        self._pending: list[int] = []
        self._flat: list[torch.Tensor] = []           # persistent reduced-precision-free buckets (param dtype)
        self._wire: list[torch.Tensor | None] = []    # bf16 wire buffers (compress) or None
        self._work: list = []
        self._launched: set[int] = set()
        self._seen: set[int] = set()
        self._staged: list[list] = []                 # bucket -> [(gradient or None, bucket slice)] waiting for _flush()
        self._views: dict[int, tuple] = {}            # param index -> (bucket, its slice shaped like the parameter, fused staging?)
        self.stats = {"wait_s": 0.0, "steps": 0}
        self.timing = False                            # True: finish() brackets its waits with events on the compute stream
        self._wait_events: list = []
        self._bucket_events: list = []                 # timing: per step {bucket: (launch event, wait start, wait end)}
        self._step_launch: dict = {}
        self._cap = None                               # begin_capture() .. end_capture(): the hooks run inside a graph capture
        self._comm_stream = None                       # replay_launch(): the stream the collectives of a replayed step start from
        if self.active and self.world > 1 and self.params and self.params[0].is_cuda and dist.get_backend(group) == "nccl":
            # RCCL's kernels and the persistent conv kernels share the chip: a launch script that did not size both gets a
            # warning (the run is correct, the overlap is not), not an error -- bench.py --gpus N sets both
            try:
                check_cu_budget(cus_per_gpu=torch.cuda.get_device_properties(self.params[0].device).multi_processor_count)
            except RuntimeError as e:
                import warnings

                warnings.warn(str(e), RuntimeWarning, stacklevel=2)
        if self.active and broadcast:
            with torch.no_grad():
                for p in self.params:
                    dist.broadcast(p.detach(), src=0, group=group)
                for b in module.buffers():
                    dist.broadcast(b.detach(), src=0, group=group)
            # the broadcast wrote through detached aliases: bump the version counters so that caches keyed
            # on Tensor._version (ops._packed_conv3) repack on the non-zero ranks
            torch.autograd.graph.increment_version(self.params)
        if self.active:
            order = _static_ready_order(module, self.params)
            if order is not None:
                self._build_buckets(order)
            for i, p in enumerate(self.params):
                p.register_post_accumulate_grad_hook(self._make_hook(i))

    # -- bucket construction ------------------------------------------------------------
    def _build_buckets(self, order: list[int]):
        seen, dedup = set(), []
        for i in order:  # de-duplicate (a parameter used twice fires once per backward, but be safe)
            if i not in seen:
                seen.add(i)
                dedup.append(i)
        order = dedup + [i for i in range(len(self.params)) if i not in seen]  # unused params last
        buckets, cur, cur_bytes, cur_key = [], [], 0, None
        for i in order:
            p = self.params[i]
            key = (p.dtype, p.device)
            n = p.numel() * p.element_size()
            if cur and (cur_bytes + n > self.bucket_bytes or key != cur_key):
                buckets.append(cur)
                cur, cur_bytes = [], 0
            cur.append(i)
            cur_bytes += n
            cur_key = key
        if cur:
            buckets.append(cur)
        self._buckets = buckets
        self._bucket_of = {i: b for b, idxs in enumerate(buckets) for i in idxs}
        self._pending = [len(b) for b in buckets]
        self._flat, self._wire, self._slot = [], [], {}
        for b, idxs in enumerate(buckets):
            p0 = self.params[idxs[0]]
            align = max(1, 16 // p0.element_size())  # every slice starts on a 16-byte boundary (fused optimiser)
            off = 0
            for i in idxs:
                self._slot[i] = (b, off)
                n = self.params[i].numel()
                off += n + (-n) % align
            self._flat.append(torch.zeros(off, dtype=p0.dtype, device=p0.device))
            self._wire.append(torch.zeros(off, dtype=torch.bfloat16, device=p0.device) if self.compress == "bf16" else None)
        self._staged = [[] for _ in buckets]
        self._views = {}
        for i, (b, off) in self._slot.items():  # the slices never move: built once, not per hook call
            p = self.params[i]
            self._views[i] = (b, self._flat[b][off : off + p.numel()].view_as(p), p.is_cuda and p.dtype == torch.float32)

    def bucket_layout(self):
        """[(n_params, bytes)] per bucket, in launch order (None before the buckets exist)."""
        if self._buckets is None:
            return None
        return [(len(b), sum(self.params[i].numel() * self.params[i].element_size() for i in b)) for b in self._buckets]

    # -- hooks ----------------------------------------------------------------------------
    def _make_hook(self, i: int):
        def hook(param):
            if not self.enabled:
                return
            if i in self._seen:
                raise RuntimeError("BucketedDataParallel: a parameter received a second gradient before finish() "
                                   "(one backward per finish(); gradient accumulation is not supported)")
            self._seen.add(i)
            if self._buckets is None:
                self._order.append(i)  # no static order: the first pass only records it
                return
            self._stage(i)
            b = self._bucket_of[i]
            self._pending[b] -= 1
            if self._pending[b] == 0:
                self._launch(b)

        return hook

    def _stage(self, i: int):
        """g / world -> the parameter's slice of its bucket; p.grad becomes that slice.  On the GPU the copy is deferred: the
        hook only notes (gradient, slice), and _flush() moves a whole bucket in one launch of tdx_stage_scaled when its last
        gradient has arrived (139 five-microsecond launches per step otherwise); elsewhere one torch.mul per parameter."""
        p = self.params[i]
        b, view, fusable = self._views[i]
        g = p.grad
        if fusable and (g is None or (g.dtype == torch.float32 and g.is_contiguous() and g.numel() == view.numel()
                                      and g.device == view.device)):
            self._staged[b].append((g, view))  # holds g until the bucket's launch
        elif g is None:
            view.zero_()
        elif g.data_ptr() != view.data_ptr():
            torch.mul(g, 1.0 / self.world, out=view)
        else:  # already the view (grad kept from the last step and accumulated in place): scale in place
            view.mul_(1.0 / self.world)
        p.grad = view

    def _flush(self, b: int):
        """The staged gradients of bucket b -> its flat buffer, scaled by 1 / world, on the current stream (one launch per 64)."""
        items = self._staged[b]
        if not items:
            return
        self._staged[b] = []
        from . import _lib as L

        tab = (L.StageItem * len(items))()
        for k, (g, view) in enumerate(items):
            tab[k].src = None if g is None else g.data_ptr()
            tab[k].dst = view.data_ptr()
            tab[k].n = view.numel()
        L.call("tdx_stage_scaled", tab, len(items), 1.0 / self.world, L.stream())

    def _launch(self, b: int):
        assert b not in self._launched, "bucket launched twice in one step"
        self._flush(b)
        if self._cap is not None:
            # inside a graph capture: no collective can be captured (gloo runs on the host; this runtime has no event-record
            # graph nodes to release an RCCL stream either).  Leave a mark the HOST can see behind the bucket's staging
            # kernels instead; replay_launch() starts the all-reduce when the mark appears
            from . import _lib as L

            cap = self._cap
            cap["order"].append(b)
            L.call("tdx_signal_host", cap["flag"].data_ptr(), L.ptr(cap["gen"]), len(cap["order"]), L.stream())
            self._launched.add(b)
            return
        flat = self._flat[b]
        if self.compress == "bf16":
            wire = self._wire[b]
            wire.copy_(flat)
            self._work.append((dist.all_reduce(wire, op=dist.ReduceOp.SUM, group=self.group, async_op=True), b))
        else:
            self._work.append((dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True), b))
        self._launched.add(b)
        if self.timing and flat.is_cuda:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()  # on the compute stream: where in backward this bucket's collective was enqueued
            self._step_launch[b] = ev

    # -- captured training step (training.GraphedTrainingStep) -------------------------------
    def begin_capture(self, gen: torch.Tensor):
        """The next backward pass runs inside a hipGraph capture: the hooks' staging kernels (g / world -> bucket slice) are
        captured as they are; where a hook would launch a bucket's all-reduce it captures `tdx_signal_host` instead (a word
        of pinned host memory <- gen * 64 + position in the launch order; `gen`: int32 device scalar the graph increments
        once per replay).  Needs the buckets (a static ready order, or one eager step before)."""
        if not self.active:
            return
        if self._buckets is None:
            raise RuntimeError("BucketedDataParallel.begin_capture: run one eager step first (the ready order is not known yet)")
        assert not self._seen and not self._launched, "begin_capture() in the middle of a step"
        flag = torch.zeros(16, dtype=torch.int32).pin_memory()
        self._cap = {"gen": gen, "flag": flag, "order": []}

    def end_capture(self):
        """-> the plan replay_launch() needs for every replay of the graph just captured (None when inactive)."""
        if self._cap is None:
            return None
        cap, self._cap = self._cap, None
        per_bucket = {}
        for i in self._seen:
            per_bucket[self._bucket_of[i]] = per_bucket.get(self._bucket_of[i], 0) + 1
        plan = {"order": list(cap["order"]), "flag": cap["flag"], "flag_np": cap["flag"].numpy(), "params": frozenset(self._seen),
                "per_bucket": per_bucket}
        self._launched.clear()
        self._seen.clear()
        self._pending = [len(b) for b in self._buckets]
        return plan

    def replay_launch(self, plan, generation: int, timeout_s: float = 120.0):
        """After `graph.replay()` of a step captured between begin_capture() / end_capture(): account for the gradients the
        graph stages, and start every bucket's all-reduce from the communication stream the moment its mark shows up in
        host memory -- i.e. while the rest of the captured backward is still running.  `generation`: how many times the graph
        has been replayed, this replay included.  finish() completes the step as usual."""
        if plan is None or not self.active or not self.enabled:
            return
        dev = self._flat[0].device
        if dev.type == "cuda" and self._comm_stream is None:
            self._comm_stream = torch.cuda.Stream(device=dev)
        self._seen |= plan["params"]
        for b, n in plan["per_bucket"].items():
            self._pending[b] -= n
        flag, base = plan["flag_np"], (generation * 64) & 0x7FFFFFFF
        for k, b in enumerate(plan["order"], 1):
            if k == len(plan["order"]) and self._comm_stream is not None:
                # the last bucket fills up where the captured backward ends: no polling -- its all-reduce is ordered behind the
                # whole replay by a stream dependency, and the host runs ahead into finish(), the optimiser and the next replay
                # instead of sitting out the tail of backward (the gap that left between two steps: 0.5-0.7 ms)
                self._comm_stream.wait_stream(torch.cuda.current_stream(dev))
            else:
                deadline = time.perf_counter() + timeout_s
                while int(flag[0]) < base + k:  # written by the captured tdx_signal_host behind bucket b's staging kernels
                    if time.perf_counter() > deadline:
                        raise RuntimeError(f"replay_launch: bucket {b} of the captured step never signalled (flag {int(flag[0])}, "
                                           f"expected >= {base + k})")
            if self._comm_stream is not None:
                with torch.cuda.stream(self._comm_stream):
                    self._launch(b)
            else:
                self._launch(b)

    # -- per-step API ----------------------------------------------------------------------
    def finish(self):
        """Complete the step's gradient exchange.  Must be called after backward()."""
        if not self.active or not self.enabled:
            self._seen.clear()
            return
        if self._buckets is None:
            # no static order and this was the first backward: build the buckets in the observed order and
            # reduce them now (no overlap on this step only)
            self._build_buckets(self._order)
            for i in range(len(self.params)):
                if i in self._seen:
                    self._stage(i)
        for b, idxs in enumerate(self._buckets):
            if b not in self._launched:
                for i in idxs:  # parameters that received no gradient this step contribute zeros
                    if i not in self._seen:
                        self._stage(i)
                self._launch(b)
        t0 = time.perf_counter()
        timed = self.timing and self._flat and self._flat[0].is_cuda
        if timed:
            # what backward did not hide: between these two events the compute stream only waits for the collectives
            # (and widens the bf16 wire buffers); everything enqueued before e0 is backward work
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        per_bucket = {}
        for w, b in self._work:
            if timed:
                w0 = torch.cuda.Event(enable_timing=True)
                w0.record()
            w.wait()
            if self.compress == "bf16":
                self._flat[b].copy_(self._wire[b])
            if timed:
                w1 = torch.cuda.Event(enable_timing=True)
                w1.record()
                per_bucket[b] = (self._step_launch.get(b), w0, w1)
        if timed:
            e1.record()
            self._wait_events.append((e0, e1))
            self._bucket_events.append(per_bucket)
        self._step_launch = {}
        self.stats["wait_s"] += time.perf_counter() - t0
        self.stats["steps"] += 1
        self._work.clear()
        self._launched.clear()
        self._seen.clear()
        self._pending = [len(b) for b in self._buckets]

    def exposed_ms(self) -> list:
        """Per-step GPU time the compute stream spent waiting for the gradient all-reduces inside finish() since the
        last call (needs ``timing = True``; synchronises)."""
        torch.cuda.synchronize()
        out = [a.elapsed_time(b) for a, b in self._wait_events]
        self._wait_events.clear()
        return out

    def bucket_report(self, backward_start_events=None) -> list:
        """Per bucket, averaged over the timed steps since the last call: what finish() still had to wait for it on the
        compute stream (`exposed_ms`) and, with the events bench.py records in front of every backward(), how far into
        backward its all-reduce was enqueued (`enqueued_at_ms`).  Needs ``timing = True``; synchronises."""
        torch.cuda.synchronize()
        steps, self._bucket_events = self._bucket_events, []
        if not steps or self._buckets is None:
            return []
        layout = self.bucket_layout()
        out = []
        for b in range(len(self._buckets)):
            exposed, at = [], []
            for k, per in enumerate(steps):
                if b not in per:
                    continue
                launch, w0, w1 = per[b]
                exposed.append(w0.elapsed_time(w1))
                if launch is not None and backward_start_events is not None and k < len(backward_start_events):
                    at.append(backward_start_events[k].elapsed_time(launch))
            out.append({"bucket": b, "params": layout[b][0], "bytes": layout[b][1],
                        "exposed_ms": sum(exposed) / max(len(exposed), 1),
                        "enqueued_at_ms": (sum(at) / len(at)) if at else None})
        return out

    def allreduce_alone_ms(self, repeats: int = 3) -> list:
        """Each bucket's all-reduce by itself (nothing else on the GPU), ms, averaged over `repeats`: the denominator
        of a per-bucket hidden fraction."""
        if not self.active or self._buckets is None:
            return []
        out = []
        for b in range(len(self._buckets)):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(repeats):
                self._launch(b)
                for w, _b in self._work:
                    w.wait()
                self._work.clear()
                self._launched.clear()
            torch.cuda.synchronize()
            out.append(1e3 * (time.perf_counter() - t0) / repeats)
        self._step_launch = {}
        return out

    def allreduce_only(self):
        """Launch and wait for one all-reduce per bucket on whatever the buckets hold (bench.py's
        communication-only leg: the time the collectives take when nothing overlaps them)."""
        if not self.active or self._buckets is None:
            return
        for b in range(len(self._buckets)):
            self._launch(b)
        for w, b in self._work:
            w.wait()
        self._work.clear()
        self._launched.clear()


def shard_trajectories(n_total: int, rank: int, world: int) -> range:
    """Contiguous block of trajectory ids owned by `rank` (sampling is embarrassingly
    parallel: no collective inside the loop)."""
    per, extra = divmod(n_total, world)
    start = rank * per + min(rank, extra)
    return range(start, start + per + (1 if rank < extra else 0))
