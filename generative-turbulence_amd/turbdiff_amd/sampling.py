"""hipGraph-captured DDPM ancestral sampler (BASELINE.json configs[3]).

The reference's loop (ddpm.py:767-816) pays per step: a host->device scalar for t, ~12 small
elementwise kernels, two index scatters and the Python/launch overhead of ~400 kernels of the
U-Net.  Here ONE reverse step -- U-Net forward, the two Philox noise draws, the fused update
and the on-device decrement of t -- is captured once into a hipGraph and replayed T times;
nothing in the step touches the host.  The timestep and the RNG offset live in device memory,
so a replay needs no new arguments.

Noise comes from a counter-based generator keyed by (seed, trajectory id, offset): results do
not depend on how trajectories are sharded over GPUs.
"""

from __future__ import annotations

import torch

from . import ops


class GraphSampler:
    def __init__(self, diffusion, x_bcs, C, cell_idx, seed: int = 0, trajectory_ids=None, use_graph: bool = True):
        self.d = diffusion
        self.x_bcs = x_bcs.contiguous().float()
        self.C = C
        B = self.x_bcs.shape[0]
        dev = self.x_bcs.device
        self.B, self.dev, self.seed = B, dev, int(seed)
        V = self.x_bcs[0, 0].numel()
        self.mask, _ = diffusion.domain_mask(cell_idx, V)
        ids = list(range(B)) if trajectory_ids is None else list(trajectory_ids)
        assert len(ids) == B
        self.stream_ids = torch.tensor(ids, dtype=torch.int64, device=dev)
        self.offset = torch.zeros(1, dtype=torch.int64, device=dev)
        self.t = torch.zeros(1, dtype=torch.int64, device=dev)
        self.x_t = torch.empty_like(self.x_bcs)
        self.z = torch.empty_like(self.x_bcs)
        self.z2 = torch.empty_like(self.x_bcs)
        self.use_graph = use_graph
        self.graph = None
        with torch.no_grad():
            self.enc = diffusion.model.encode_local(C)
        self.reset()

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
        eps = d.model(self.x_t, self.t.expand(self.B), self.C, **kw)
        self._randn(self.z)
        if d.noise_bcs:
            self._randn(self.z2)
        ops.p_sample_step(self.x_t, eps, self.z, self.z2 if d.noise_bcs else None, self.x_bcs, self.mask, d.step_tables,
                          d.num_timesteps, self.t, d.noise_bcs, d.clip_denoised, out=self.x_t)
        self.t.sub_(1)

    @torch.no_grad()
    def _capture(self):
        # warm up on a side stream (packs weights, sizes the allocator), then capture
        state = (self.x_t.clone(), self.t.clone(), self.offset.clone())
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            self._step()
        torch.cuda.current_stream().wait_stream(s)
        self.graph = torch.cuda.CUDAGraph()
        # capture on the warm-up stream: the conv kernels' scratch arena is bound per stream (_lib.ensure_scratch), the
        # warm-up step created this stream's, so nothing is allocated inside the capture and the replayed graph never
        # shares an arena with eager work on another stream
        self._capture_stream = s
        with torch.cuda.graph(self.graph, stream=s):
            self._step()
        self.x_t.copy_(state[0]); self.t.copy_(state[1]); self.offset.copy_(state[2])
        # The graph has the addresses of the packed weight operands (ops._pack_cache) baked in.  Keep those
        # buffers alive for as long as the graph lives, and remember the parameter versions they were packed
        # from: a weight update (optimiser step, load_state_dict) makes replay() re-capture instead of running on
        # stale operands.
        self._held = list(ops._pack_cache.values())
        self._versions = [(p, p._version) for p in self.d.parameters()]

    def _stale(self) -> bool:
        return any(p._version != v for p, v in self._versions)

    @torch.no_grad()
    def run_steps(self, n: int):
        n = min(n, self.steps_left)
        if self.use_graph and (self.graph is None or self._stale()):
            self._capture()
        for _ in range(n):
            if self.use_graph:
                self.graph.replay()
            else:
                self._step()
        self.steps_left -= n
        return self.x_t

    @torch.no_grad()
    def sample(self, start_from: int | None = None):
        """Full trajectory; returns the (B, F, X, Y, Z) sample with BC cells set to x_bcs."""
        self.reset(start_from)
        self.run_steps(self.steps_left)
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
