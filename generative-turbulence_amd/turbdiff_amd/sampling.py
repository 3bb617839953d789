"""hipGraph-captured DDPM ancestral sampler (BASELINE.json configs[3]).

The reference's loop (ddpm.py:767-816) pays per step: a host->device scalar for t, ~12 small
elementwise kernels, two index scatters and the Python/launch overhead of ~400 kernels of the
U-Net.  Here ONE reverse step -- U-Net forward, then one kernel that draws the two Philox
noises where they are consumed and applies the update, then the on-device advance of the RNG
offset and decrement of t -- is captured once into a hipGraph and replayed T times;
nothing in the step touches the host.  The timestep and the RNG offset live in device memory,
so a replay needs no new arguments.

Noise comes from a counter-based generator keyed by (seed, trajectory id, offset): results do
not depend on how trajectories are sharded over GPUs.
"""

from __future__ import annotations


import torch

from . import _lib as L
from . import ops

# module constants (tests flip them; they were environment A/B switches while the two routes were being compared):
# FUSED_STEP_NOISE = False: draw z / z2 into tensors with tdx_randn_batched and run tdx_p_sample_step on them (same values);
# COND_TABLE = False: the time MLP runs every reverse step instead of one look-up in a per-timestep table
FUSED_STEP_NOISE = True
COND_TABLE = True


class GraphSampler:
    """One captured reverse step, replayed T times.  Everything the captured step reads lives in buffers the sampler
    OWNS (boundary values, conditioning tensors, in-domain mask, cached conditioning conv), so `rebind` can point an
    existing graph at another batch / geometry of the same shape by copying into them -- `GaussianDiffusion.
    p_sample_loop` keeps one sampler per input shape and re-uses its graph from call to call."""

    def __init__(self, diffusion, x_bcs, C, cell_idx, seed: int = 0, trajectory_ids=None, use_graph: bool = True, nonce: int = 0,
                 capture_stream=None):
        self.d = diffusion
        self.x_bcs = x_bcs.detach().float().contiguous().clone()
        self.C = {k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in C.items()} if isinstance(C, dict) else C
        B = self.x_bcs.shape[0]
        dev = self.x_bcs.device
        self.B, self.dev, self.seed = B, dev, int(seed)
        V = self.x_bcs[0, 0].numel()
        self.mask = diffusion.domain_mask(cell_idx, V)[0].clone()
        ids = list(range(B)) if trajectory_ids is None else list(trajectory_ids)
        assert len(ids) == B
        # Philox key = `seed` (a launch argument: fixed once captured); counter words = (offset + element, stream id);
        # stream id = trajectory id in the low 32 bits, `nonce` in the high 31: the nonce lives in device memory, so a
        # captured graph draws fresh noise per call without a re-capture (p_sample_loop passes one per call)
        self.stream_ids = torch.tensor(self._sids(ids, nonce), dtype=torch.int64, device=dev)
        self.offset = torch.zeros(1, dtype=torch.int64, device=dev)
        self.t = torch.zeros(1, dtype=torch.int64, device=dev)
        self.x_t = torch.empty_like(self.x_bcs)
        # noise drawn inside the update kernel (tdx_p_sample_step_rng) unless the layout rules that out; only then do the
        # two noise tensors exist at all
        self.fused_noise = FUSED_STEP_NOISE and ops.p_sample_step_rng_supported(self.x_t)
        self.z = None if self.fused_noise else torch.empty_like(self.x_bcs)
        self.z2 = None if self.fused_noise or not diffusion.noise_bcs else torch.empty_like(self.x_bcs)
        self.use_graph = use_graph
        self.graph = None
        # ONE stream for every capture of this sampler (its scratch arena is per stream); samplers of one diffusion share it
        self._capture_stream = capture_stream
        with torch.no_grad():
            self.enc = diffusion.model.encode_local(self.C)
        self.c_table = None  # (T, c_dim) conditioning vectors per timestep (DenoisingModel.conditioning_table) or None
        self._table_versions = None
        self._refresh_tables()
        self.reset()

    @staticmethod
    def _sids(ids, nonce):
        assert 0 <= int(nonce) < 2**31 and all(0 <= int(i) < 2**32 for i in ids)
        return [(int(nonce) << 32) | int(i) for i in ids]

    @staticmethod
    def signature_of(diffusion, x_bcs, C):
        """What must match for `rebind`: shapes / dtypes of everything the captured step reads, and the switches the
        captured step was built under."""
        c = tuple(sorted((str(k), tuple(v.shape), str(v.dtype)) for k, v in C.items() if torch.is_tensor(v))) \
            if isinstance(C, dict) else None
        m = diffusion.model
        return (tuple(x_bcs.shape), str(x_bcs.device), c, getattr(m, "compute_dtype", None), getattr(m, "conv_impl", None),
                L.conv_impl(), diffusion.noise_bcs, diffusion.clip_denoised, diffusion.num_timesteps)

    def signature(self):
        return self.signature_of(self.d, self.x_bcs, self.C)

    @torch.no_grad()
    def rebind(self, x_bcs, C, cell_idx, nonce=None, trajectory_ids=None):
        """Another batch (and possibly another geometry) of the SAME shapes: copy it into the buffers the captured graph
        reads.  The cached conditioning conv is recomputed into its existing tensors."""
        self.x_bcs.copy_(x_bcs)
        if isinstance(self.C, dict):
            for k, v in C.items():
                if torch.is_tensor(v):
                    self.C[k].copy_(v)
        self.mask.copy_(self.d.domain_mask(cell_idx, self.x_bcs[0, 0].numel())[0])
        self._refresh_enc()
        if nonce is not None or trajectory_ids is not None:
            ids = [int(i) & 0xFFFFFFFF for i in self.stream_ids.tolist()] if trajectory_ids is None else list(trajectory_ids)
            nn = int(self.stream_ids[0].item()) >> 32 if nonce is None else nonce
            self.stream_ids.copy_(torch.tensor(self._sids(ids, nn), dtype=torch.int64))
        return self

    @torch.no_grad()
    def _refresh_enc(self):
        """encode_local(C) (and the cached conditioning conv hanging off it) again, into the tensors the captured graph
        reads: after another geometry was bound or the weights changed."""
        if self.enc is None:
            return
        enc = self.d.model.encode_local(self.C)
        old, new = getattr(self.enc, "first_conv_partial", None), getattr(enc, "first_conv_partial", None)
        if (old is None) != (new is None) or enc.shape != self.enc.shape or enc.dtype != self.enc.dtype:
            self.enc, self.graph = enc, None  # another structure than the captured one: capture again on these tensors
        else:
            self.enc.copy_(enc)
            if old is not None:
                old[1].copy_(new[1])

    @torch.no_grad()
    def _refresh_tables(self):
        """What the step reads that is a function of the WEIGHTS (and the bound geometry) alone: recomputed into the same
        tensors after a weight update (the captured graph keeps their addresses)."""
        if getattr(self, "_table_versions", None) is not None:  # not the constructor's call: enc was just made there
            self._refresh_enc()
        tab = self.d.model.conditioning_table(self.C, self.d.num_timesteps) if COND_TABLE else None
        old = self.c_table
        if tab is not None and old is not None and tab.shape == old.shape and tab.dtype == old.dtype:
            old.copy_(tab)
        else:
            if tab is not None or old is not None:
                self.graph = None  # another structure than the captured one
            self.c_table = tab
        self._table_versions = [(p, p._version) for p in self.d.parameters()]

    def _changed(self, seen) -> bool:
        """Have the diffusion's parameters changed since `seen` = [(tensor, version)] was recorded -- updated in place
        (optimiser step, load_state_dict) or replaced by other tensors (another model assigned, .to(device))?"""
        cur = list(self.d.parameters())
        return len(cur) != len(seen) or any(p is not q or p._version != v for p, (q, v) in zip(cur, seen))

    def _tables_stale(self) -> bool:
        return self._changed(self._table_versions)

    # ---- state
    def _randn(self, out):
        return ops.randn_philox_batched(out, self.seed, self.stream_ids, self.offset)

    @torch.no_grad()
    def reset(self, start_from: int | None = None):
        """x_T ~ N(0, I) (or q_sample(x_bcs, start_from - 1)); t <- T - 1; RNG offset <- 0."""
        d = self.d
        self.offset.zero_()
        T = d.num_timesteps if start_from is None else start_from
        self.t.fill_(T - 1)
        self._randn(self.x_t)
        if start_from is not None:
            tt = torch.full((1,), start_from - 1, dtype=torch.int64, device=self.dev)
            self.x_t.copy_(ops.q_sample(self.x_bcs, self.x_t, d.sqrt_alphas_cumprod, d.sqrt_one_minus_alphas_cumprod, tt))
        if not d.noise_bcs:
            self.x_t.copy_(torch.where(self.mask.view(self.x_t.shape[-3:]).bool(), self.x_t, self.x_bcs))
        self.steps_left = T

    # ---- one reverse step (graph body)
    def _step(self):
        d = self.d
        kw = {"encoded_local": self.enc} if self.enc is not None else {}
        if self.c_table is not None:
            # all trajectories are at the same t: one row of the table instead of the time MLP (~12 launches per step)
            kw["cond"] = self.c_table.index_select(0, self.t).expand(self.B, -1)
        eps = d.model(self.x_t, self.t.expand(self.B), self.C, **kw)
        if self.fused_noise:
            # same draws, same counters, bit-identical x_{t-1}; also advances the offset and decrements t
            ops.p_sample_step_rng(self.x_t, eps, self.x_bcs, self.mask, d.step_tables, d.num_timesteps, self.t,
                                  d.noise_bcs, d.clip_denoised, self.seed, self.stream_ids, self.offset, out=self.x_t)
            return
        self._randn(self.z)
        if d.noise_bcs:
            self._randn(self.z2)
        ops.p_sample_step(self.x_t, eps, self.z, self.z2 if d.noise_bcs else None, self.x_bcs, self.mask, d.step_tables,
                          d.num_timesteps, self.t, d.noise_bcs, d.clip_denoised, out=self.x_t)
        self.t.sub_(1)

    @torch.no_grad()
    def _capture(self):
        # warm up on the sampler's own side stream (packs weights, sizes the allocator, creates that stream's scratch
        # arena), then capture on the same stream: the conv kernels' arena is bound per stream (_lib.ensure_scratch), so
        # nothing is allocated inside the capture and the replayed graph never shares an arena with eager work on another
        # stream.  Re-captures (after a weight update) re-use the stream -- and with it the arena.
        state = (self.x_t.clone(), self.t.clone(), self.offset.clone())
        if self._capture_stream is None:
            self._capture_stream = torch.cuda.Stream(device=self.dev)
        s = self._capture_stream
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            self._step()
            self._arena = L.scratch_arena(self.dev)  # held for as long as the graph lives (the arena map is bounded)
        torch.cuda.current_stream().wait_stream(s)
        self.graph = torch.cuda.CUDAGraph()
        # capture_error_mode "thread_local": only THIS thread's calls are checked against the capture.  In a process with an RCCL
        # communicator the watchdog thread of ProcessGroupNCCL polls the events of recent collectives (a barrier just before the
        # sampling leg is enough); under the default "global" mode its hipEventQuery during our capture raises
        # hipErrorStreamCaptureUnsupported and aborts the process (seen in 2 of 6 runs of the captured data-parallel step)
        with torch.cuda.graph(self.graph, stream=s, capture_error_mode="thread_local"):
            self._step()
        self.x_t.copy_(state[0]); self.t.copy_(state[1]); self.offset.copy_(state[2])
        # The graph has the addresses of the packed weight operands (ops._pack_cache) baked in.  Keep those
        # buffers alive for as long as the graph lives, and remember the parameter versions they were packed
        # from: a weight update (optimiser step, load_state_dict) makes replay() re-capture instead of running on
        # stale operands.
        self._held = list(ops._pack_cache.values())
        self._versions = [(p, p._version) for p in self.d.parameters()]

    def _stale(self) -> bool:
        return self._changed(self._versions)

    @torch.no_grad()
    def run_steps(self, n: int, pbar: bool = False):
        n = min(n, self.steps_left)
        if n <= 0:
            return self.x_t  # finished trajectory: t is -1, a warm-up / capture step would index the tables out of range
        if self._tables_stale():
            self._refresh_tables()
        if self.use_graph and (self.graph is None or self._stale()):
            self._capture()
        steps = range(n)
        if pbar:
            from tqdm.auto import tqdm

            steps = tqdm(steps, desc="sampling loop time step", total=n, position=1)
        for _ in steps:
            if self.use_graph:
                self.graph.replay()
            else:
                self._step()
        self.steps_left -= n
        return self.x_t

    @torch.no_grad()
    def sample(self, start_from: int | None = None, pbar: bool = False):
        """Full trajectory; returns the (B, F, X, Y, Z) sample with BC cells set to x_bcs."""
        self.reset(start_from)
        self.run_steps(self.steps_left, pbar=pbar)
        return self.x_t.clone()

    # ---- the same noise, as tensors, for cross-checks against the eager loop
    @torch.no_grad()
    def noise_stream(self):
        """Generator yielding exactly the noise tensors sample() consumes, in order."""
        off = torch.zeros(1, dtype=torch.int64, device=self.dev)
        while True:
            out = torch.empty_like(self.x_bcs)
            ops.randn_philox_batched(out, self.seed, self.stream_ids, off)
            yield out
