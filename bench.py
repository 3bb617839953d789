#!/usr/bin/env python3
"""Benchmark of the turbdiff denoising-diffusion hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one DDPM training step of the full turbdiff U-Net (dim 32, 4 levels, GroupNorm(8),
c_local 4; 55.2 M parameters) on one synthetic batch of 192x64x48 (u, p) volumes:
GaussianDiffusion.forward (t ~ randint, q_sample, U-Net forward, masked l2 loss) + backward
+ gradient all-reduce (N > 1) + clip-by-norm 0.1 + RAdam -- BASELINE.json configs[1] / [2].
Inputs are resident in HBM before the timed region.  One JSON line is printed by rank 0:
  value        whole-job voxels/s = N * B * 192*64*48 * K / (max over ranks of the K-step time)
  roofline     the dominant kernel (MFMA implicit-GEMM 3x3x3 conv, forward launches): algorithmic
               FLOPs of all its launches / their HIP-event time, against the dense MFMA peak of the mode
  cpu_baseline the CPU oracle (a PyTorch-CPU port of the reference math, fp32) timed on the host cores on
               B = 1 of the same workload, at all cores and at 8 threads (N = 1, rank 0 only; runs FIRST,
               so that the GPU legs are the last thing the driver's utilisation sampler sees)
  extra        accuracy   rel-L2 of every mode's eps-hat against the CPU oracle at 192x64x48 (same weights, inputs)
               parity_modes  the same step in the modes that meet north_star's 1e-4 gate (f32s: fp32 tensors +
                             split-precision convs; f32: IEEE fp32 MFMA convs), each with its own roofline block
               fwd-only and sampling legs (DDPM samples/s: >= 50 reverse steps timed; the full T = 1000 loop
               when --steps >= 20; the public p_sample_loop: graph vs eager, and the reference's B = 1 protocol);
               cfg1: BASELINE configs[0] (48x32x32, 2 levels, T = 10) on the GPU and on the CPU oracle;
               real_grid: the dataset's 194x50x50 grid at the reference batch (bf16, f32s);
               N > 1: overlap report of the gradient all-reduce and the sharded
               sampling leg of BASELINE configs[3] (8 trajectories per GPU)
"""

import argparse
import json
import os
import sys
import time
from pathlib import Path
from types import SimpleNamespace

ROOT = Path(__file__).resolve().parent
for p in (ROOT, ROOT / "generative-turbulence_amd"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))

import torch  # noqa: E402

GRID = (192, 64, 48)
V = GRID[0] * GRID[1] * GRID[2]
PEAK_BF16_TFLOPS = 2500.0  # dense, MI355X_MICROARCH.md
PEAK_F32_TFLOPS = 157.3
# f32s issues three bf16 MFMAs per algorithmic product: its matrix-core ceiling is a third of the bf16 peak
# fp16 MFMA forms cost the same matrix-pipe cycles as the bf16 ones on gfx950: one peak for both 16-bit modes
PEAK = {"bf16": PEAK_BF16_TFLOPS, "fp16": PEAK_BF16_TFLOPS, "f32": PEAK_F32_TFLOPS, "f32s": PEAK_BF16_TFLOPS / 3}
CONV_BYTES = {"bf16": 1417.6e6, "fp16": 1417.6e6, "f32": 2835.2e6, "f32s": 2835.2e6}  # conv-only bytes per sample forward
KERNEL = {"bf16": "conv3_mfma_kernel", "fp16": "conv3_mfma_kernel", "f32": "conv3_mfma_f32_kernel", "f32s": "conv3_mfma_split_kernel"}
MODE_DTYPE = {"bf16": torch.bfloat16, "fp16": torch.float16, "f32": torch.float32, "f32s": torch.float32}
LOSS_ELEMENTS = lambda B, cell_idx: B * 4 * int(cell_idx.numel())  # values behind the loss's mean (sizes the fp16 loss scale)
ACCURACY_T = 250  # diffusion time of the accuracy probe


def synthetic_inputs(B, device, grid=GRID):
    """x ~ N(0,1) seed 1234, cell-type conditioning seed 1235, interior box
    minus a 12x16x32 obstacle at x in [13, 25)."""
    W, H, D = grid
    x = torch.randn(B, 4, W, H, D, generator=torch.Generator().manual_seed(1234))
    c = torch.randn(4, W, H, D, generator=torch.Generator().manual_seed(1235))
    m = torch.zeros(W, H, D, dtype=torch.bool)
    m[1:-1, 1:-1, 1:-1] = True
    y0, z0 = (H - 16) // 2, 0
    m[13:25, y0 : y0 + 16, z0 : z0 + 32] = False
    cell_idx = torch.nonzero(m.flatten()).squeeze(-1)
    return x.to(device), c.to(device), cell_idx.to(device)


def new_denoiser(timesteps=500):
    from turbdiff_amd.models.ddpm import DenoisingModel

    torch.manual_seed(0)
    return DenoisingModel(in_features=4, out_features=4, c_local_features=4, c_global_features=0, timesteps=timesteps,
                          dim=32, u_net_levels=4, norm_type="group")


def build_model(device, dtype=None, timesteps=500):
    """The BASELINE configs[1] diffusion; dtype (torch.bfloat16 / torch.float32) selects the compute dtype right
    away (tools/*.py), None leaves it to set_mode."""
    from turbdiff_amd.models.ddpm import GaussianDiffusion

    diff = GaussianDiffusion(new_denoiser(timesteps), timesteps=timesteps, beta_schedule="log-snr-linear", loss_type="l2",
                             noise_bcs=True)
    diff = diff.to(device)
    if dtype is not None:
        diff.model.set_compute_dtype(dtype)
    return diff


def set_mode(diff, mode):
    """bf16: bf16 storage + bf16 MFMA; fp16: fp16 storage + fp16 MFMA (TF32's 11 significand bits at the bf16 kernels' speed;
    training steps run under ClipRAdam's loss scale); f32: fp32 storage, IEEE fp32 MFMA convs; f32s: fp32 storage,
    split-precision convs (bf16 hi + lo, three MFMAs per product)."""
    diff.model.set_compute_dtype(MODE_DTYPE[mode])
    diff.model.conv_impl = "split" if mode == "f32s" else "auto"  # per model: no process-wide switch, no environment edits


def cpu_baseline(budget_s=120.0):
    """fwd+bwd voxels/s of the CPU oracle on B = 1 of the same workload at 8 threads (3 warm-up + 5 timed iterations,
    median: BASELINE.md §3) and at all host threads (1 + 2; fewer if the budget runs out).  Also returns the oracle's eps-hat of the
    accuracy probe (the forward of those iterations) for extra.accuracy."""
    from oracle import turbdiff_oracle as O

    net = new_denoiser()
    sd = {k: v.clone().requires_grad_() for k, v in net.state_dict().items()}
    del net
    x, c, cell_idx = synthetic_inputs(1, "cpu")
    buf = O.schedule_buffers("log-snr-linear", 500)
    t = torch.tensor([ACCURACY_T])
    noise = torch.randn(x.shape, generator=torch.Generator().manual_seed(1))
    all_threads = torch.get_num_threads()
    legs, eps_ref = [], None
    for nthreads in sorted({all_threads, min(8, all_threads)}, reverse=True):
        torch.set_num_threads(nthreads)
        # BASELINE.md section 3's protocol (3 warm-up + 5 timed, median) at 8 threads -- the leg the survey's probe and
        # every comparison refer to; the all-threads leg (oversubscribed: slower) is bounded to 1 + 2 iterations
        warm, timed = (3, 5) if nthreads <= 8 else (1, 2)
        times, t_start = [], time.perf_counter()
        for it in range(warm + timed):
            t0 = time.perf_counter()
            loss, eps_hat = O.p_losses(sd, buf, x, t, c, cell_idx, noise, timesteps=500, noise_bcs=True)
            torch.autograd.grad(loss, list(sd.values()))
            dt = time.perf_counter() - t0
            eps_ref = eps_hat.detach()
            if it >= warm:
                times.append(dt)
            if time.perf_counter() - t_start > budget_s * 0.6 and times:
                break
        med = sorted(times)[len(times) // 2]
        legs.append({"threads": nthreads, "voxels_per_s": V / med, "s_per_iteration": med, "timed_iterations": len(times),
                     "warmup_iterations": warm})
    torch.set_num_threads(all_threads)
    head = max(legs, key=lambda l: l["voxels_per_s"])  # the better of the two thread counts is the baseline
    out = {"value": head["voxels_per_s"], "unit": "voxels/s", "cores": head["threads"], "kind": "port",
           "sample": f"B=1 fwd+bwd of the same 192x64x48 step, fp32, median of {head['timed_iterations']} after "
                     f"{head['warmup_iterations']} warm-up ({head['s_per_iteration']:.2f} s each)",
           "legs": legs}
    return out, {"x_t": O.q_sample(buf, x, t, noise), "t": t, "eps": eps_ref}


def torch_rocm_baseline(B, dev, amp, steps=2):
    """The reference's arithmetic path as it runs on this GPU through stock PyTorch-ROCm ops
    (MIOpen conv3d, ATen group_norm / interpolate / SDPA): the oracle's functional restatement on
    `cuda`, fp32 weights, optionally under bf16 autocast, same step (fwd + bwd + clip + RAdam).
    A reported baseline (--torch-baseline), never part of `value`."""
    from oracle import turbdiff_oracle as O

    net = new_denoiser()
    sd = {k: v.clone().to(dev).requires_grad_() for k, v in net.state_dict().items()}
    del net
    params = list(sd.values())
    opt = torch.optim.RAdam(params, lr=1e-4)
    x, c, cell_idx = synthetic_inputs(B, dev)
    buf = {k: v.to(dev) for k, v in O.schedule_buffers("log-snr-linear", 500).items()}
    noise = torch.randn(x.shape, device=dev)

    def step():
        t = torch.randint(0, 500, (B,), device=dev)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
            loss, _ = O.p_losses(sd, buf, x, t, c, cell_idx, noise, timesteps=500, noise_bcs=True)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(params, 0.1)
        opt.step()
        opt.zero_grad(set_to_none=True)

    step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    res = {"ms_per_step": ms, "voxels_per_s": B * V / (ms * 1e-3), "batch": B, "precision": "bf16 autocast" if amp else "fp32",
           "peak_mem_GB": torch.cuda.max_memory_allocated() / 1e9}
    del sd, params, opt
    torch.cuda.empty_cache()
    return res


# forward instantiations of the matrix-core conv kernels in a rocprofv3 kernel name (ZERO_PAD / ZP template argument false)
FWD_KERNEL_PATTERNS = {
    # <NT, ZP, LW, HF, Z4> | <NT, XT, ZERO_PAD, PERM, EXT, HF>; HF = fp16 operands, Z4 = 4-deep bricks
    "bf16": r"conv3_ring_kernel<\d, false, \d, false, (true|false)>|conv3_mfma_kernel<\d, (true|false), false, (true|false), (true|false), false>",
    "fp16": r"conv3_ring_kernel<\d, false, \d, true, (true|false)>|conv3_mfma_kernel<\d, (true|false), false, (true|false), (true|false), true>",
    "f32": r"conv3_mfma_f32_kernel<\d, false,", "f32s": r"conv3_mfma_split_kernel<\d, false,"}


FWD_KERNEL_FAMILY = r"conv3_ring_kernel<|conv3_mfma_kernel<|conv3_mfma_f32_kernel<|conv3_mfma_split_kernel<"


def measured_traffic(mode="bf16"):
    """HBM bytes per launch of the mode's forward brick / ring conv kernels from the committed rocprofv3 PMC passes
    (profiles/*_traffic.json, produced by tools/collect_profiles.sh + summarize_profiles.py at B = 6): the newest file that
    holds launches of those kernels AND was collected on the conv kernels of this tree (its `kernel_sources_sha16` equals
    the fingerprint of csrc/tdx_conv*; a file that describes older kernels is refused and named under `refused`).
    -> (bytes per launch | None, {file, git_head, ...}, launches sampled).  Same launches as
    roofline.algorithmic_bytes_per_launch: the forward conv calls served by the brick / ring kernels (the deep levels'
    small-grid launches are a different kernel)."""
    import re

    from turbdiff_amd._lib import kernel_sources_fingerprint

    pat = FWD_KERNEL_PATTERNS[mode]
    now = kernel_sources_fingerprint()
    refused = []
    # rNN tags: newest round first; within a round the set collected with the kernels one after the other before the
    # one with the side stream on (`*ss_traffic.json`; the forward launches have nothing beside them either way)
    files = sorted((ROOT / "profiles").glob("*_traffic.json"), key=lambda f: (f.name[:3], not f.name.endswith("ss_traffic.json"), f.name),
                   reverse=True)
    for f in files:
        data = json.loads(f.read_text())
        num = den = 0.0
        for name, k in data["kernels"].items():
            if re.search(pat, name):
                n = k["launches_sampled"]
                num += n * (k["read_bytes_per_launch"] + k["write_bytes_per_launch"])
                den += n
        if not den:
            # a file from before the kernels carried the operand format in their names (rounds <= 5): other kernels
            if data.get("kernel_sources_sha16") != now and any(re.search(FWD_KERNEL_FAMILY, name) for name in data["kernels"]):
                refused.append({"file": f"profiles/{f.name}", "git_head": data.get("git_head"),
                                "why": "collected on other conv-kernel sources than this tree's"})
            continue
        if data.get("kernel_sources_sha16") != now:
            refused.append({"file": f"profiles/{f.name}", "git_head": data.get("git_head"),
                            "why": "collected on other conv-kernel sources than this tree's"})
            continue
        return num / den, {"file": f"profiles/{f.name}", "git_head": data.get("git_head"),
                           "kernel_sources_sha16": now, "side_stream": data.get("side_stream")}, int(den)
    return None, {"file": None, "kernel_sources_sha16": now, "refused": refused[:3]}, 0


def live_traffic(mode, batch, timeout_s=240):
    """HBM bytes per launch of the mode's forward brick / ring conv kernels, measured NOW on this box: two child runs of this
    script (3 steps of the headline step, nothing else) under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and `... WRITE_SIZE`
    -- separate passes, counters only, as MI355X_MICROARCH.md prescribes (FETCH_SIZE counts 64 B per 128-B request on gfx950:
    doubled; both are in KB).  Started from main() BEFORE this process touches the GPU; children are ordinary subprocesses
    (`rocprofv3 ... -- python3 bench.py ...`).  -> (bytes per launch | None, source dict, launches sampled)."""
    import csv
    import re
    import shutil
    import subprocess
    import tempfile

    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if Path("/opt/rocm/bin/rocprofv3").exists() else None)
    if exe is None:
        return None, {"live": False, "why": "rocprofv3 not on PATH"}, 0
    pat = FWD_KERNEL_PATTERNS[mode]
    env = dict(os.environ, TDX_BENCH_CHILD="1", TMPDIR="/tmp", TDX_WGRAD_STREAM="0")
    sums, n_launch, note = {}, {}, []
    with tempfile.TemporaryDirectory(dir="/tmp", prefix="tdx_traffic_") as tmp:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = Path(tmp) / counter
            cmd = [exe, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", str(out), "--", sys.executable,
                   str(Path(__file__).resolve()), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extra", "--dtype", mode,
                   "--batch", str(batch)]
            try:
                r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout_s)
            except subprocess.TimeoutExpired:
                return None, {"live": False, "why": f"rocprofv3 --pmc {counter} pass did not finish in {timeout_s} s"}, 0
            files = list(out.glob("**/*counter_collection.csv"))
            if r.returncode != 0 or not files:
                return None, {"live": False, "why": f"rocprofv3 --pmc {counter} pass failed (rc {r.returncode}): {r.stderr[-200:]}"}, 0
            tot = cnt = 0.0
            for row in csv.DictReader(open(files[0])):
                if row["Counter_Name"] == counter and re.search(pat, row["Kernel_Name"]):
                    tot += float(row["Counter_Value"])
                    cnt += 1
            sums[counter], n_launch[counter] = tot, cnt
    if not n_launch["FETCH_SIZE"] or n_launch["FETCH_SIZE"] != n_launch["WRITE_SIZE"]:
        return None, {"live": False, "why": f"no matching launches in the counter tables ({n_launch})"}, 0
    n = n_launch["FETCH_SIZE"]
    per_launch = (2.0 * sums["FETCH_SIZE"] + sums["WRITE_SIZE"]) * 1024.0 / n
    return per_launch, {"live": True, "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (two passes of 3 headline steps in "
                        "child processes of this run, kernels one after the other: TDX_WGRAD_STREAM=0); FETCH_SIZE x 2 (gfx950), KB -> bytes",
                        "read_bytes_per_launch": 2.0 * sums["FETCH_SIZE"] * 1024.0 / n, "write_bytes_per_launch": sums["WRITE_SIZE"] * 1024.0 / n}, int(n)


CONV_CALLS = {"tdx_conv3_fwd", "tdx_conv3_fwd_gn", "tdx_conv3_bwd_data", "tdx_conv3_bwd_data_add", "tdx_conv3_bwd_weight"}
# The headline's timed region carries HIP events only around the launches its roofline block is about (22 calls a step); the
# per-call table (`kernels`, events around all ~66 conv calls) and, at N > 1, the events around backward() and the all-reduce
# waits come from a second, instrumented leg of the same K steps (their cost is within the leg-to-leg noise of +-0.2 ms on
# one GPU; the point is that the headline region of a data-parallel run holds no per-bucket instrumentation)
FWD_CALLS = {"tdx_conv3_fwd", "tdx_conv3_fwd_gn"}


def merged_kernel_times(timer, where=None):
    kern = timer.summary(where)
    # tdx_conv3_fwd_gn = the same conv kernel with the GroupNorm statistics in its epilogue
    # (+ an 8 us finalize kernel inside the bracket): count it as a forward launch
    for alias, name in (("tdx_conv3_fwd_gn", "tdx_conv3_fwd"), ("tdx_conv3_bwd_data_add", "tdx_conv3_bwd_data")):
        if alias in kern:
            f = kern.setdefault(name, {"launches": 0, "ms": 0.0, "work": 0.0, "bytes": 0.0})
            g = kern.pop(alias)
            for k in f:
                f[k] += g[k]
    return kern


LIVE_TRAFFIC = {}  # mode -> live_traffic() result of this run (filled by main() before the GPU is touched)


def roofline_block(timer, mode, B, K):
    """The dominant kernel = the matrix-core 3x3x3 conv, forward launches.  achieved / frac / traffic /
    algorithmic_bytes_per_launch all refer to the SAME launches: the tdx_conv3_fwd(_gn) calls served by the brick and
    ring kernels (on the shipped model 16 of the 22 forward convs of a step; the 6 deep-level launches run the
    small-grid kernel and are reported, together with everything, under all_forward_launches)."""
    from turbdiff_amd import _lib

    main = merged_kernel_times(timer, lambda m: m.get("kind") in (_lib.KERNEL_BRICK, _lib.KERNEL_RING)).get("tdx_conv3_fwd")
    every = merged_kernel_times(timer).get("tdx_conv3_fwd")
    if not main or main["ms"] <= 0:
        main = every
    if not main or main["ms"] <= 0:
        return None
    ach = main["work"] / (main["ms"] * 1e-3) / 1e12
    ach_all = every["work"] / (every["ms"] * 1e-3) / 1e12
    traffic, tsrc, tn = LIVE_TRAFFIC.get(mode) or (None, None, 0)
    if traffic is None:  # no live measurement (no rocprofv3, N > 1, --no-live-traffic): the committed PMC file of this tree's kernels
        why = tsrc
        traffic, tsrc, tn = measured_traffic(mode)
        if why is not None:
            tsrc = dict(tsrc, live_attempt=why)
    ring = merged_kernel_times(timer, lambda m: m.get("kind") == _lib.KERNEL_RING).get("tdx_conv3_fwd")
    return {"bound": "mfma", "achieved": ach, "peak": PEAK[mode], "unit": "TFLOP/s", "frac": ach / PEAK[mode],
            "traffic": traffic, "traffic_unit": "HBM bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE), same launches",
            "traffic_source": tsrc, "traffic_launches_sampled": tn,
            "algorithmic_bytes_per_launch": main["bytes"] / main["launches"] if main["bytes"] else None,
            "kernel": ("conv3_ring_kernel + conv3_mfma_kernel" if mode in ("bf16", "fp16") else KERNEL[mode]) +
                      " (tdx_conv3_fwd launches on the brick / ring kernels)",
            "launches": main["launches"], "avg_launch_ms": main["ms"] / main["launches"],
            "ring_kernel": ({"launches": ring["launches"], "achieved": ring["work"] / (ring["ms"] * 1e-3) / 1e12,
                             "frac": ring["work"] / (ring["ms"] * 1e-3) / 1e12 / PEAK[mode]} if ring and ring["ms"] > 0 else None),
            "all_forward_launches": {"launches": every["launches"], "achieved": ach_all, "frac": ach_all / PEAK[mode],
                                     "ms_per_step": every["ms"] / K},
            "conv_bandwidth_roofline_frac": (CONV_BYTES[mode] * B * K / (every["ms"] * 1e-3)) / 8e12}


def kernel_table(kern, K):
    return {k: {"launches": v["launches"], "ms_per_step": v["ms"] / K, "tflops": v["work"] / max(v["ms"], 1e-9) / 1e9}
            for k, v in kern.items()}


def new_optimizer(diff, mode, n_loss_elements):
    """ClipRAdam (clip 0.1 + RAdam); in fp16 with the loss scale the trainer would choose for this batch."""
    from turbdiff_amd.optim import ClipRAdam
    from turbdiff_amd.training import DiffusionTrainer

    scale = DiffusionTrainer.initial_loss_scale(n_loss_elements) if mode == "fp16" else None
    return ClipRAdam(diff.parameters(), lr=1e-4, max_norm=0.1, loss_scale=scale)


def timed_train_steps(diff, x, C, md, mode, steps, warmup):
    """ms per training step (fwd + bwd + clip 0.1 + RAdam) of `diff` on (x, C, md) in `mode`, single rank."""
    set_mode(diff, mode)
    opt = new_optimizer(diff, mode, LOSS_ELEMENTS(x.shape[0], md.cell_idx))

    def step():
        loss, _ = diff(x, C, md, None)
        opt.scale_loss(loss).backward()
        opt.step()
        opt.zero_grad(set_to_none=True)

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / steps


class _Task:
    """What training.GraphedTrainingStep asks of a task, around a bare GaussianDiffusion and dense inputs."""

    def __init__(self, diff, opt=None):
        self.model, self._opt = diff, opt  # (_opt: GraphedTrainingStep reads the loss scale from it)

    def _model_input(self, b):
        return b.x, b.C

    def _cell_idx(self, b):
        return b.cell_idx

    def parameters(self):
        return self.model.parameters()


def host_and_graph_step(diff, x, C, md, mode, steps, warmup):
    """The same training step eagerly and with forward + backward replayed from ONE captured hipGraph
    (training.GraphedTrainingStep; clip + RAdam stay eager): ms per step and the HOST's share of it (time to enqueue a
    step, no device sync inside the timed loop; the device is drained before and after)."""
    from turbdiff_amd.training import GraphedTrainingStep

    set_mode(diff, mode)
    batch = SimpleNamespace(x=x, C=C, cell_idx=md.cell_idx)
    res = {}
    for kind in ("eager", "graph"):
        diff.zero_grad(set_to_none=True)
        opt = new_optimizer(diff, mode, LOSS_ELEMENTS(x.shape[0], md.cell_idx))
        gs = GraphedTrainingStep(_Task(diff, opt)) if kind == "graph" else None

        def step():
            if gs is not None:
                gs(batch)
            else:
                opt.zero_grad(set_to_none=True)
                loss, _ = diff(x, C, md, None)
                opt.scale_loss(loss).backward()
                del loss
            opt.step()

        for _ in range(warmup):
            step()
        torch.cuda.synchronize()
        t0, c0 = time.perf_counter(), time.process_time()
        for _ in range(steps):
            step()
        t1, c1 = time.perf_counter(), time.process_time()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        # host_cpu: CPU time of the process (issuing thread + autograd's backward thread: what the host WORKS per step; a
        # hipGraphLaunch that waits for room in the launch queue spins, so on a device-bound step it counts the wait);
        # host_enqueue: wall time until the last step is issued
        res[kind] = {"ms_per_step": 1e3 * (t2 - t0) / steps, "host_enqueue_ms_per_step": 1e3 * (t1 - t0) / steps,
                     "host_cpu_ms_per_step": 1e3 * (c1 - c0) / steps}
        del gs, opt
        diff.zero_grad(set_to_none=True)
    res["note"] = ("graph = forward + backward as one captured hipGraph per input signature (t and noise drawn inside, weights "
                   "re-packed inside); clip + RAdam eager in both; host_enqueue = wall time to issue the steps")
    return res


CFG1_GRID = (48, 32, 32)


def new_cfg1_denoiser():
    """BASELINE configs[0] / SURVEY 8(d) cfg1: 2-level U-Net, dim 32, T = 10."""
    from turbdiff_amd.models.ddpm import DenoisingModel

    torch.manual_seed(0)
    return DenoisingModel(in_features=4, out_features=4, c_local_features=4, c_global_features=0, timesteps=10, dim=32,
                          u_net_levels=2, norm_type="group")


def cfg1_leg(dev, with_cpu):
    """north_star's second volume: one 48x32x32x4 synthetic volume through the 2-level U-Net -- training step and the
    10-step DDPM loop (the public p_sample_loop) per mode on the GPU, and the CPU oracle (the reference's own
    CPU-runnable case) beside it."""
    from turbdiff_amd.models.conditioning import Conditioning
    from turbdiff_amd.models.ddpm import GaussianDiffusion

    W, H, D = CFG1_GRID
    v1 = W * H * D
    res = {"workload": "BASELINE configs[0]: one 48x32x32 (u,p) volume, 2-level U-Net dim 32 GN(8), T = 10, B = 1", "modes": {}}
    diff = GaussianDiffusion(new_cfg1_denoiser(), timesteps=10, beta_schedule="log-snr-linear", loss_type="l2",
                             noise_bcs=True).to(dev)
    x, c, cell_idx = synthetic_inputs(1, dev, CFG1_GRID)
    C, md = {Conditioning.Type.CELL_TYPE: c}, SimpleNamespace(cell_idx=cell_idx)
    sd0 = {k: v.detach().cpu().clone() for k, v in diff.model.state_dict().items()}
    for mode in ("bf16", "fp16", "f32s", "f32"):
        diff.model.load_state_dict(sd0)
        ms = timed_train_steps(diff, x, C, md, mode, 20, 5)
        diff.p_sample_loop(x, C, cell_idx)  # capture
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            diff.p_sample_loop(x, C, cell_idx)
        torch.cuda.synchronize()
        loop_s = (time.perf_counter() - t0) / 5
        res["modes"][mode] = {"ms_per_step": ms, "voxels_per_s": v1 / (ms * 1e-3), "sample_loop_ms": 1e3 * loop_s,
                              "ddpm_samples_per_s_T10": 1.0 / loop_s}
        hg = host_and_graph_step(diff, x, C, md, mode, 20, 5)
        res["modes"][mode]["captured_step"] = hg
        res["modes"][mode]["voxels_per_s_captured_step"] = v1 / (hg["graph"]["ms_per_step"] * 1e-3)
    if with_cpu:
        from oracle import turbdiff_oracle as O

        nth = torch.get_num_threads()
        torch.set_num_threads(min(8, nth))
        sd = {k: v.clone().requires_grad_() for k, v in sd0.items()}
        buf = O.schedule_buffers("log-snr-linear", 10)
        xc, cc, ic = x.cpu(), c.cpu(), cell_idx.cpu()
        noise = torch.randn(xc.shape, generator=torch.Generator().manual_seed(1))
        t = torch.tensor([5])
        times = []
        for it in range(4):
            t0 = time.perf_counter()
            loss, _ = O.p_losses(sd, buf, xc, t, cc, ic, noise, timesteps=10, noise_bcs=True)
            torch.autograd.grad(loss, list(sd.values()))
            if it:
                times.append(time.perf_counter() - t0)
        med = sorted(times)[len(times) // 2]
        gen = torch.Generator().manual_seed(2)
        with torch.no_grad():
            sdn = {k: v.detach() for k, v in sd.items()}
            t0 = time.perf_counter()
            O.p_sample_loop(sdn, buf, xc, cc, ic, (torch.randn(xc.shape, generator=gen) for _ in range(64)), timesteps=10,
                            noise_bcs=True)
            loop = time.perf_counter() - t0
        torch.set_num_threads(nth)
        res["cpu_oracle"] = {"threads": min(8, nth), "s_per_step_fwd_bwd": med, "voxels_per_s": v1 / med,
                             "sample_loop_s": loop, "ddpm_samples_per_s_T10": 1.0 / loop,
                             "note": "CPU oracle (PyTorch-CPU port of the reference math, fp32): median of 3 fwd+bwd after 1 "
                                     "warm-up; one 10-step loop"}
    return res


def cfg5_attention_leg(dev):
    """BASELINE configs[4]: the bottleneck attention core (reference attention.py:9-15, ddpm.py:295-308; heads = 4,
    dim_head = 32) at 96x32x24 = 73 728 tokens, B = 1, on the MFMA flash kernels with fp16 and bf16 operands: forward and
    backward ms by HIP events on the launching stream, algorithmic TFLOP/s (forward 4 N^2 d h; backward 2.5x that:
    dP, dV, dQ, dK and the recomputed S) against the 2.5 PFLOP/s dense peak, and rel-L2 of 128 sampled query rows against
    the CPU oracle's SDPA on the same rounded inputs."""
    from turbdiff_amd import ops

    B, H, D, N = 1, 4, 32, 96 * 32 * 24
    flops = 4.0 * N * N * D * H * B
    res = {"workload": "BASELINE configs[4]: attention core, N = 96x32x24 = 73 728 tokens, 4 heads x 32, B = 1",
           "fwd_flops": flops, "bwd_flops": 2.5 * flops, "peak_tflops": 2500.0,
           "algorithmic_bytes": 4.0 * N * H * D * 2 * B, "modes": {}}
    gen = torch.Generator(device=dev).manual_seed(11)
    base = torch.randn(B, N, 3 * H * D, device=dev, generator=gen)
    rows = torch.randperm(N, generator=torch.Generator().manual_seed(12))[:128]
    for name, dt in (("fp16", torch.float16), ("bf16", torch.bfloat16)):
        qkv = base.to(dt)
        for _ in range(2):
            out = ops.attention(qkv, H)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 5
        s.record()
        for _ in range(n):
            out = ops.attention(qkv, H)
        e.record()
        torch.cuda.synchronize()
        fwd = s.elapsed_time(e) / n
        qg = qkv.clone().requires_grad_()
        o2 = ops.attention(qg, H)
        go = torch.randn(o2.shape, device=dev, generator=gen).to(dt)
        for _ in range(2):
            o2.backward(go, retain_graph=True)
        s.record()
        for _ in range(n):
            o2.backward(go, retain_graph=True)
        e.record()
        torch.cuda.synchronize()
        bwd = s.elapsed_time(e) / n
        m = {"fwd_ms": fwd, "bwd_ms": bwd, "fwd_tflops": flops / fwd / 1e9, "bwd_tflops": 2.5 * flops / bwd / 1e9,
             "fwd_frac_of_mfma_peak": flops / fwd / 1e9 / 2500.0, "bwd_frac_of_mfma_peak": 2.5 * flops / bwd / 1e9 / 2500.0,
             "hbm_frac_fwd": 4.0 * N * H * D * 2 * B / (fwd * 1e-3) / 8e12}
        try:
            from oracle import turbdiff_oracle as O  # the checker, on 128 query rows

            q, k, v = qkv.float().cpu().reshape(B, N, 3, H, D).permute(2, 0, 3, 1, 4)
            ref = O.sdpa(q[:, :, rows], k, v)
            got = out.float().cpu().reshape(B, N, H, D)[:, rows].permute(0, 2, 1, 3)
            m["rel_l2_vs_oracle_128_rows"] = ((got - ref).norm() / ref.norm()).item()
        except Exception as ex:  # noqa: BLE001
            m["rel_l2_vs_oracle_128_rows"] = f"{type(ex).__name__}: {ex}"[:200]
        res["modes"][name] = m
        del qkv, qg, o2, go, out
    return res


REAL_GRID = (194, 50, 50)  # the dataset's grid (reference scripts/grid-embedding.py:69)


def real_grid_leg(dev, B):
    """The full 4-level dim-32 U-Net on the reference's real grid 194x50x50 (every axis ragged against the 8x8x8 bricks,
    every level down to 24x6x6 ragged): ms per training step at the reference batch in bf16 and f32s."""
    from turbdiff_amd.models.conditioning import Conditioning

    diff = build_model(dev)
    x, c, cell_idx = synthetic_inputs(B, dev, REAL_GRID)
    C, md = {Conditioning.Type.CELL_TYPE: c}, SimpleNamespace(cell_idx=cell_idx)
    vr = REAL_GRID[0] * REAL_GRID[1] * REAL_GRID[2]
    res = {"grid": list(REAL_GRID), "batch": B, "modes": {}}
    for mode, k, w in (("bf16", 10, 3), ("fp16", 10, 3), ("f32s", 5, 2)):
        ms = timed_train_steps(diff, x, C, md, mode, k, w)
        res["modes"][mode] = {"ms_per_step": ms, "voxels_per_s": B * vr / (ms * 1e-3), "steps": k}
    return res


def deterministic_leg(diff, x, C, md, mode, B, cell_idx):
    """TDX_DETERMINISTIC=1 (csrc/tdx_ordered.hip): time per training step with every fp32 atomic merge of the backward pass on its
    ordered route, and whether two identical backward passes leave the same bits in every parameter gradient -- with the
    switch and, for comparison, without it."""
    def two_passes():
        t = torch.full((B,), ACCURACY_T, device=x.device, dtype=torch.long)
        noise = torch.randn(x.shape, device=x.device, generator=torch.Generator(device=x.device).manual_seed(1))
        S = 2.0 ** 12 if mode == "fp16" else 1.0
        runs = []
        for _ in range(2):
            diff.zero_grad(set_to_none=True)
            loss, _ = diff.p_losses(x, t, C, md, None, noise=noise)
            (loss * S).backward()
            torch.cuda.synchronize()
            runs.append({n: p.grad.clone() for n, p in diff.model.named_parameters()})
        diff.zero_grad(set_to_none=True)
        return sum(not torch.equal(runs[0][n], runs[1][n]) for n in runs[0]), len(runs[0])

    set_mode(diff, mode)
    differ_default, n = two_passes()
    before = os.environ.get("TDX_DETERMINISTIC")
    os.environ["TDX_DETERMINISTIC"] = "1"  # read per call by the library and by _lib.call
    try:
        differ_det, _ = two_passes()
        ms = timed_train_steps(diff, x, C, md, mode, 10, 3)
    finally:
        if before is None:
            del os.environ["TDX_DETERMINISTIC"]
        else:
            os.environ["TDX_DETERMINISTIC"] = before
    return {"ms_per_step": ms, "parameter_gradients": n, "differing_between_two_runs": differ_det,
            "differing_between_two_runs_by_default": differ_default,
            "note": "TDX_DETERMINISTIC=1: ordered merges instead of fp32 atomics in the backward pass; two identical backward passes compared bit for bit"}


def _parse_cpulist(text):
    out = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        out += list(range(int(lo), int(hi or lo) + 1))
    return out


def pin_host_threads(local_rank, local_world):
    """Before anything touches the GPU: bind this rank's host threads (the issuing thread, autograd's backward thread, the
    collective's proxy thread) to cores of its GPU's NUMA node -- eight ranks x two busy threads on one host otherwise
    migrate across sockets and stretch the enqueue path the step depends on when per-GPU batches are small.  The GPU's
    node comes from sysfs (AMD display / accelerator-class PCI functions in bus order = HIP's default device order; `numa_node` /
    `local_cpulist`), without a HIP call; the node's cores are split evenly among the ranks that share it.  Where sysfs
    says nothing (a container without it, one NUMA node) the allowed cores are split evenly by rank.  Returns what was
    done, for extra.overlap.host_affinity."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except AttributeError:
        return {"pinned": False, "why": "no sched_setaffinity on this platform"}
    if os.environ.get("TDX_BENCH_PIN", "1") == "0" or local_world <= 1 or len(allowed) < 2 * local_world:
        return {"pinned": False, "why": "TDX_BENCH_PIN=0" if os.environ.get("TDX_BENCH_PIN", "1") == "0" else
                f"{len(allowed)} allowed cores for {local_world} ranks"}
    cores, source, node = None, "even split of the allowed cores by rank", None
    try:
        gpus = []
        for dev in sorted(Path("/sys/bus/pci/devices").iterdir()):
            # AMD GPUs: display controllers (0x03....) or, for the Instinct parts, processing accelerators (0x12....)
            if (dev / "vendor").read_text().strip() == "0x1002" and (dev / "class").read_text().strip()[:4] in ("0x03", "0x12"):
                gpus.append(dev)
        if len(gpus) >= local_world:
            mine = gpus[local_rank]
            node = int((mine / "numa_node").read_text())
            local = [c for c in _parse_cpulist((mine / "local_cpulist").read_text()) if c in allowed]
            sharers = [i for i, g in enumerate(gpus[:local_world]) if int((g / "numa_node").read_text()) == node]
            per = len(local) // max(len(sharers), 1)
            if node >= 0 and per >= 2:
                k = sharers.index(local_rank)
                cores, source = local[k * per : (k + 1) * per], f"sysfs: {mine.name} on NUMA node {node}, shared by ranks {sharers}"
    except (OSError, ValueError):
        pass
    if not cores:
        per = len(allowed) // local_world
        cores = allowed[local_rank * per : (local_rank + 1) * per]
    os.sched_setaffinity(0, cores)
    return {"pinned": True, "cores": [cores[0], cores[-1]], "n_cores": len(cores), "numa_node": node, "source": source}


def self_launch(n):
    """`python -m torch.distributed.run --nnodes=1 --nproc-per-node n ... bench.py <same flags>` as a CHILD process
    on a free loopback port; returns its exit code."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL's intra-node transport on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--torch-find", choices=("fast", "full"), default="fast",
                    help="MIOpen find mode of the stock PyTorch-ROCm leg: fast (default, seconds) or full (exhaustive, ~10 min cold)")
    ap.add_argument("--steps", type=int, default=20)   # 0.5 s of timed steps; 5 steps after 2 warm-up measured 3 % slower
    ap.add_argument("--warmup", type=int, default=5)   # (clocks and allocator not settled) than 20 after 5 on the same box
    ap.add_argument("--batch", type=int, default=6, help="per-GPU batch (reference: 6, config/model/diffusion.yaml:3)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16", "f32", "f32s"],
                    help="mode of the headline line.  bf16: bf16 storage + bf16 MFMA (BASELINE configs[1]); fp16: fp16 storage + fp16 "
                         "MFMA under a loss scale (the reference's TF32-grade arithmetic, same kernels); f32: fp32 storage, "
                         "IEEE fp32 MFMA convs; f32s: fp32 storage, split-precision convs (bf16 hi + lo, 3 MFMAs per product). "
                         "The other two modes are reported under extra.parity_modes / extra.accuracy in the same run")
    ap.add_argument("--sample-steps", type=int, default=50, help="reverse steps timed for the sampling leg (0 = skip)")
    ap.add_argument("--sample-batch", type=int, default=8)
    ap.add_argument("--optimizer", default="fused", choices=["fused", "torch"],
                    help="fused: ClipRAdam (clip + RAdam in 3 launches); torch: clip_grad_norm_ + torch.optim.RAdam")
    ap.add_argument("--compress", default=None, choices=[None, "bf16"], help="N > 1: gradients travel as bf16")
    ap.add_argument("--persistent-cus", type=int, default=224,
                    help="N > 1: CUs the persistent conv kernels occupy (TDX_PERSISTENT_CUS; the rest is left to RCCL)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true")
    ap.add_argument("--no-parity-modes", action="store_true", help="skip the f32s / f32 legs of extra.parity_modes")
    ap.add_argument("--no-torch-baseline", action="store_true",
                    help="skip the stock PyTorch-ROCm leg (BASELINE configs[1]: 'HIP Conv3d vs PyTorch-ROCm Conv3d': the same "
                         "step through MIOpen conv3d / ATen ops, 2 steps bf16 autocast + 1 step fp32 -> extra.torch_rocm_reference)")
    ap.add_argument("--torch-baseline", action="store_true", help="(default since round 3; kept for old command lines)")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not measure roofline.traffic in this run (two rocprofv3 --pmc child passes, ~1 min); quote the committed "
                         "PMC file of this tree's kernels instead")
    args = ap.parse_args()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC (RCCL's intra-node transport on this driver): before HIP starts

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # bare `python bench.py --gpus N`: start the N ranks ourselves (one process per GPU under
        # torch.distributed.run) BEFORE anything in this process touches the GPU, relay rank 0's line, exit with
        # the launcher's code.  Nothing is exec'd from a process that holds a GPU context.
        sys.exit(self_launch(args.gpus))

    if int(os.environ.get("WORLD_SIZE", "1")) > 1 or os.environ.get("TDX_BENCH_FORCE_DDP") == "1":
        # Data-parallel runs: RCCL's kernels and the persistent conv kernels must fit the chip together.  The conv kernels
        # take 224 CUs (TDX_PERSISTENT_CUS, below); RCCL is held to at most 32 channels (= workgroups, one CU each) unless
        # the caller chose otherwise -- 221 MB of gradients per ~22 ms step need ~10 GB/s per rank, far below what 32
        # channels move.  Both must be in the environment before the communicator exists.
        os.environ.setdefault("NCCL_MAX_NCHANNELS", "32")

    # roofline.traffic of the headline mode, measured on THIS box in this run: before anything here touches the GPU
    if (args.gpus == 1 and not args.no_live_traffic and not args.no_extra and os.environ.get("TDX_BENCH_CHILD") != "1"
            and "WORLD_SIZE" not in os.environ and os.environ.get("TDX_BENCH_FORCE_DDP") != "1"):
        t_live = time.perf_counter()
        live_modes = [args.dtype] + ([m for m in ("fp16", "bf16", "f32s", "f32") if m != args.dtype] if not args.no_parity_modes else [])
        deadline = t_live + 300.0  # all passes together (normally ~30 s); a pass that fails or stalls ends the attempt for every mode
        for m in live_modes:  # the headline mode and every parity mode that gets a roofline block below
            left = deadline - time.perf_counter()
            if left < 20.0:
                LIVE_TRAFFIC[m] = (None, {"live": False, "why": "the run's budget for PMC passes was used up"}, 0)
                continue
            LIVE_TRAFFIC[m] = live_traffic(m, args.batch, timeout_s=min(120.0, left))
            if LIVE_TRAFFIC[m][0] is None:
                deadline = 0.0
        LIVE_TRAFFIC["wall_s"] = round(time.perf_counter() - t_live, 1)

    # N > 1: NUMA-local cores for this rank's host threads, before anything (the library, the communicator) touches the GPU
    affinity = pin_host_threads(int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1"))))

    from turbdiff_amd import _lib, parallel

    # TDX_BENCH_BACKEND=gloo + TDX_BENCH_ONE_DEVICE=1: all ranks on cuda:0 with gloo collectives -- only for
    # exercising the N > 1 code path on a one-GPU box (tests/test_bench_multirank.py); never a measurement
    # TDX_BENCH_FORCE_DDP=1 with one rank: the N > 1 code path (communicator, gradient hooks + asynchronous all-reduce on RCCL's
    # stream, overlap report, captured data-parallel step, sharded sampling leg) on ONE GPU over RCCL at world size 1 -- a dry
    # run of everything the multi-GPU runs execute except a second peer (tests/test_bench_multirank.py); never a measurement
    force_ddp = os.environ.get("TDX_BENCH_FORCE_DDP") == "1" and int(os.environ.get("WORLD_SIZE", "1")) == 1
    rank, world, local = parallel.init_from_env(os.environ.get("TDX_BENCH_BACKEND", "nccl"), force=force_ddp)
    if os.environ.get("TDX_BENCH_ONE_DEVICE") == "1":
        local = 0
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    multi = world > 1 or force_ddp  # the data-parallel legs run; the single-GPU extras do not
    if multi:
        # The ring conv / producer-consumer weight-gradient kernels launch one workgroup per CU and fill its registers and
        # LDS; RCCL's kernels need CUs of their own, and a persistent launch that finds some taken runs in two rounds.
        # Under data parallelism they therefore use 224 of the 256 CUs (28 per XCD) unless the caller chose a number
        # (single-GPU cost: profiles/r11_persistent_cus_ab.txt; --persistent-cus 256 restores the full chip).
        os.environ.setdefault("TDX_PERSISTENT_CUS", str(args.persistent_cus))
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    B, K, Wm = args.batch, args.steps, args.warmup

    wall, t_leg = {}, [time.perf_counter()]  # wall seconds per leg of this call -> extra.wall_s

    def leg_done(name):
        now = time.perf_counter()
        wall[name] = round(now - t_leg[0], 1)
        t_leg[0] = now

    # ---- CPU leg first (rank 0, N = 1): baseline + the oracle's eps-hat for the accuracy probe
    cpu, probe = None, None
    if rank == 0 and not multi and not args.no_cpu_baseline:
        cpu, probe = cpu_baseline()
        leg_done("cpu_baseline")

    from turbdiff_amd.models.conditioning import Conditioning
    from turbdiff_amd.optim import ClipRAdam

    diff = build_model(dev)
    x, c_local, cell_idx = synthetic_inputs(B, dev)
    C = {Conditioning.Type.CELL_TYPE: c_local}
    md = SimpleNamespace(cell_idx=cell_idx)
    fused_opt = args.optimizer == "fused"

    # ---- accuracy probe: every mode's eps-hat on the oracle's x_t, BEFORE any optimiser step changes the weights
    accuracy = None
    if probe is not None:
        accuracy = {"probe": f"eps-hat at t = {ACCURACY_T}, B = 1, 192x64x48, default init seed 0; rel-L2 vs the CPU oracle"}
        xt, tt = probe["x_t"].to(dev), probe["t"].to(dev)
        with torch.no_grad():
            for m in ("bf16", "fp16", "f32s", "f32"):
                set_mode(diff, m)
                e = diff.model(xt, tt, C).float().cpu()
                accuracy[m] = ((e - probe["eps"]).norm() / probe["eps"].norm()).item()
        del xt, tt

    def barrier():
        torch.cuda.synchronize()
        if multi:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(v):
        if multi:
            tt = torch.tensor([v], device=dev, dtype=torch.float64)
            torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
            return tt.item()
        return v

    ddp = parallel.BucketedDataParallel(diff, compress=args.compress, force=force_ddp)
    bwd_events = []  # N > 1: (start, end) events around backward() of every timed step

    def run_mode(mode, steps, warmup, calls=CONV_CALLS, ddp_timing=None):
        """`steps` timed training steps in `mode` -> (elapsed s (max over ranks), merged conv-kernel times, last loss).
        calls: the library calls bracketed with HIP events; ddp_timing: events around backward() and finish()'s waits."""
        ddp_timing = multi if ddp_timing is None else ddp_timing
        set_mode(diff, mode)
        assert fused_opt or mode != "fp16", "fp16 training needs the loss-scaling optimiser (--optimizer fused)"
        opt = (new_optimizer(diff, mode, LOSS_ELEMENTS(B, cell_idx)) if fused_opt
               else torch.optim.RAdam(diff.parameters(), lr=1e-4))
        scale_loss = getattr(opt, "scale_loss", lambda l: l)

        def train_step():
            loss, _ = diff(x, C, md, None)
            if ddp.timing:
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
                ev[0].record()
                scale_loss(loss).backward()
                ev[1].record()
                bwd_events.append(ev)
            else:
                scale_loss(loss).backward()
            ddp.finish()
            if not fused_opt:
                torch.nn.utils.clip_grad_norm_(diff.parameters(), 0.1)
            opt.step()
            opt.zero_grad(set_to_none=True)
            return loss.detach()

        for _ in range(warmup):
            train_step()
        ddp.timing = ddp_timing
        bwd_events.clear()
        ddp.exposed_ms()
        timer = _lib.KernelTimer(calls)
        _lib.TIMER = timer
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = train_step()
        run_mode.host_enqueue_ms = 1e3 * (time.perf_counter() - t0) / steps  # this rank's wall time to ISSUE a step
        barrier()
        elapsed = max_over_ranks(time.perf_counter() - t0)
        _lib.TIMER = None
        ddp.timing = False
        kern = merged_kernel_times(timer)
        if hasattr(opt, "settle"):
            opt.settle()
            run_mode.skipped = opt.skipped_steps  # fp16: steps whose gradients overflowed (0 = every timed step updated)
        return elapsed, (kern, timer), loss.item(), train_step

    leg_done("build_model_and_accuracy_probe")
    # one step of lazy initialisation (per-stream scratch arenas, zero-kept workspaces, pack plans, kernel attributes: all made at
    # first use) runs before the W warm-up steps, so that a call with --warmup 0 times K steps of the product, not its set-up
    # (257 ms for the first step against 21): config.untimed_steps says so
    elapsed, (kern, timer), last_loss, train_step = run_mode(args.dtype, K, Wm + 1, calls=FWD_CALLS, ddp_timing=False)
    headline_enqueue_ms = run_mode.host_enqueue_ms
    headline_skipped = getattr(run_mode, "skipped", None)
    leg_done("headline_steps")
    instrumented_ms = None
    if not args.no_extra:
        # the same K steps again with events around EVERY conv call and (N > 1) around backward() and the all-reduce waits
        el_i, (kern, _), _, train_step = run_mode(args.dtype, K, 1)
        instrumented_ms = 1e3 * el_i / K
        leg_done("instrumented_steps")
    value = world * B * V * K / elapsed
    out = {
        "metric": "U-Net fwd+bwd voxels/sec (DDPM training step, 192x64x48x4)",
        "value": value,
        "unit": "voxels/s",
        "n_gpus": world,
        "steps": K,
        "warmup": Wm,
        "ms_per_step": 1e3 * elapsed / K,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": args.dtype,
        "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: turbdiff U-Net dim32 x 4 levels GN(8), 192x64x48 (u,p), "
                               "DDPM train step (fwd+bwd+clip+RAdam)", "per_gpu_batch": B, "global_batch": B * world,
                   "grid": list(GRID), "timesteps": 500, "parallelism": f"dp{world}",
                   "untimed_steps": f"{Wm} warm-up (--warmup) + 1 of lazy initialisation before them",
                   "optimizer": "ClipRAdam (fused clip 0.1 + RAdam)" if fused_opt else "clip_grad_norm_ + torch.optim.RAdam",
                   # what was switched away from the defaults through the environment (INTEGRATION.md lists the switches)
                   "switches": {k: v for k, v in sorted(os.environ.items()) if k.startswith("TDX_") and not k.startswith("TDX_BENCH_")},
                   "weight_gradients_on_side_stream": os.environ.get("TDX_WGRAD_STREAM", "1") != "0"},
        "loss": last_loss,
    }
    if args.dtype == "fp16":
        out["skipped_steps"] = headline_skipped  # steps whose gradients overflowed (0 = every timed step updated)
    if rank == 0:
        rb = roofline_block(timer, args.dtype, B, K)
        if rb:
            out["roofline"] = rb
        out["kernels"] = kernel_table(kern, K)
        if os.environ.get("TDX_WGRAD_STREAM", "1") != "0":
            # the weight gradients run on a side stream beside the data-gradient chain (ops._WgradSide): the two event
            # durations cover overlapped time; TDX_WGRAD_STREAM=0 gives them one after the other
            out["kernels"]["note"] = "tdx_conv3_bwd_weight and tdx_conv3_bwd_data run concurrently: durations overlap"
        if instrumented_ms is not None:
            out["kernels"]["measured_in"] = (f"a second leg of the same {K} steps with HIP events around every conv call "
                                             f"({instrumented_ms:.2f} ms per step with them); the headline region carries events "
                                             "around the forward conv launches only (roofline)")
    extra = {}
    if accuracy is not None:
        extra["accuracy"] = accuracy
        out["rel_l2_vs_cpu_oracle"] = accuracy[args.dtype]

    # ---- N > 1: how much of the gradient all-reduce hides behind backward
    if multi and not args.no_extra:
        # HIP events of the timed steps themselves: backward's span on the compute stream, and what finish() then
        # still had to wait for (the part of the all-reduce backward did not hide)
        exposed_ev = ddp.exposed_ms()
        bwd_ms = [a.elapsed_time(b) for a, b in bwd_events]
        per_bucket = ddp.bucket_report([a for a, _ in bwd_events])
        ddp.enabled = False
        barrier()
        t0 = time.perf_counter()
        for _ in range(K):
            train_step()
        barrier()
        t_nocomm = max_over_ranks(time.perf_counter() - t0) / K
        ddp.enabled = True
        train_step()  # buckets hold gradients again
        barrier()
        t0 = time.perf_counter()
        for _ in range(K):
            ddp.allreduce_only()
        barrier()
        t_comm = max_over_ranks(time.perf_counter() - t0) / K
        t_step = elapsed / K
        per_rank = [None] * world  # every rank's host enqueue time per step and where its threads are pinned
        torch.distributed.all_gather_object(per_rank, {"rank": rank, "host_enqueue_ms_per_step": headline_enqueue_ms,
                                                       "host_affinity": affinity})
        exposed = max_over_ranks(sum(exposed_ev) / max(len(exposed_ev), 1)) * 1e-3
        extra["overlap"] = {"ms_step": 1e3 * t_step, "ms_step_instrumented": instrumented_ms,
                            "ms_step_without_allreduce": 1e3 * t_nocomm,
                            "ms_allreduce_alone": 1e3 * t_comm, "ms_exposed": 1e3 * exposed,
                            "ms_backward": sum(bwd_ms) / max(len(bwd_ms), 1),
                            "method": "HIP events on the compute stream around backward() and around finish()'s waits, "
                                      "mean over the steps of the instrumented leg (the same K steps right after the "
                                      "headline's, which carry no such events), max over ranks",
                            "hidden_fraction": (1.0 - exposed / t_comm) if t_comm > 0 else None,
                            "buckets": ddp.bucket_layout(), "compress": args.compress,
                            # per bucket: when in backward its all-reduce was enqueued, how long it takes alone, what
                            # finish() still waited for it, and the fraction backward hid
                            "per_bucket": [dict(r, alone_ms=a, hidden_fraction=(1.0 - r["exposed_ms"] / a) if a > 0 else None)
                                           for r, a in zip(per_bucket, ddp.allreduce_alone_ms())],
                            "per_rank": per_rank,
                            "persistent_cus": int(os.environ.get("TDX_PERSISTENT_CUS", "256")),
                            "nccl_max_nchannels": os.environ.get("NCCL_MAX_NCHANNELS"),
                            "payload_MB": sum(b for _, b in (ddp.bucket_layout() or [])) / 1e6}

    # ---- the modes that meet the 1e-4 parity gate, same run, same model (N = 1 only: they are not scaling legs)
    if not multi and not args.no_extra and not args.no_parity_modes:
        pm = {}
        for m in ("fp16", "bf16", "f32s", "f32"):
            if m == args.dtype:
                continue
            k2 = 20 if m in ("fp16", "bf16") else 10  # the 16-bit modes are compared with each other at the headline's step count
            el, (kn, tm), _, _ = run_mode(m, k2, 3)
            d = {"ms_per_step": 1e3 * el / k2, "voxels_per_s": B * V * k2 / el, "steps": k2,
                 "roofline": roofline_block(tm, m, B, k2), "kernels": kernel_table(kn, k2)}
            if m == "fp16":
                d["skipped_steps"] = getattr(run_mode, "skipped", None)
                d["note"] = ("fp16 tensors + fp16 MFMA operands (11 significand bits: the reference's TF32 arithmetic, train.py:144-156), "
                             "ClipRAdam loss scale; same kernels as bf16 templated on the operand format")
            if accuracy is not None:
                d["rel_l2_vs_cpu_oracle"] = accuracy[m]
            pm[m] = d
        extra["parity_modes"] = pm
        set_mode(diff, args.dtype)
        leg_done("parity_modes")

    if not args.no_extra and not multi:
        # the host's share of the headline step, eager and with forward + backward replayed from one captured graph
        extra["captured_step"] = host_and_graph_step(diff, x, C, md, args.dtype, 10, 3)
        leg_done("captured_step")

    if not args.no_extra:
        set_mode(diff, args.dtype)
        if not multi:
            # forward only (the north_star's "conv U-Net forward" figure)
            with torch.no_grad():
                tq = torch.full((B,), 250, device=dev)
                for _ in range(2):
                    diff.model(x, tq, C)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(3):
                    diff.model(x, tq, C)
                torch.cuda.synchronize()
                extra["fwd_ms_per_sample"] = 1e3 * (time.perf_counter() - t0) / 3 / B
        if args.sample_steps > 0:
            # BASELINE configs[3]: T = 1000, trajectories sharded 8 per GPU, hipGraph-captured reverse step
            from turbdiff_amd.sampling import GraphSampler

            T = 1000
            sdiff = build_model(dev, timesteps=T)
            set_mode(sdiff, args.dtype)
            Bs = args.sample_batch
            ids = list(parallel.shard_trajectories(Bs * world, rank, world))
            xs, _, _ = synthetic_inputs(Bs, dev)
            sampler = GraphSampler(sdiff, xs, C, cell_idx, seed=0, trajectory_ids=ids)
            sampler.run_steps(2)  # warm-up (includes graph capture)
            full = K >= 20
            n = (T - 2) if full else min(args.sample_steps, T - 2)
            barrier()
            t0 = time.perf_counter()
            sampler.run_steps(n)
            barrier()
            per_step = max_over_ranks(time.perf_counter() - t0) / n
            extra["sampling"] = {"ms_per_reverse_step": 1e3 * per_step, "reverse_steps_timed": n, "full_loop": full,
                                 "trajectories_per_gpu": Bs, "trajectories": Bs * world, "T": T,
                                 "ddpm_samples_per_s_T1000": Bs * world / (per_step * T),
                                 "ddpm_samples_per_s_T500": Bs * world / (per_step * 500),
                                 "note": "whole-job aggregate; per-step time x T" + ("" if full else " (extrapolated from the timed steps)")}
            del sampler
            if not multi and args.dtype != "fp16":
                # the same captured reverse step on fp16 tensors / fp16 MFMA operands: sampling needs no loss scale, so
                # this is the at-speed sampler at the reference's own (TF32-grade) precision
                set_mode(sdiff, "fp16")
                s16 = GraphSampler(sdiff, xs, C, cell_idx, seed=0, trajectory_ids=ids)
                s16.run_steps(2)
                n16 = min(n, 300)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                s16.run_steps(n16)
                torch.cuda.synchronize()
                p16 = (time.perf_counter() - t0) / n16
                extra["sampling"]["fp16"] = {"ms_per_reverse_step": 1e3 * p16, "reverse_steps_timed": n16,
                                             "ddpm_samples_per_s_T1000": Bs / (p16 * T)}
                del s16
                set_mode(sdiff, args.dtype)
            if not multi:
                # the PUBLIC method (what DiffusionTraining.sample / eval_ckpt.py / a dropin user call, reference
                # diffusion.py:152-158): graph path vs the eager loop (TDX_GRAPH_SAMPLER=0), 30 reverse steps each via
                # start_from; and the reference's own timing protocol (scripts/evaluate-runtime.py:54-96: B = 1, sync,
                # wall clock around one full sample()) on the full T = 1000 loop
                import turbdiff_amd.models.ddpm as D

                def public_loop(xb, start):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    sdiff.p_sample_loop(xb, C, cell_idx, start_from=start)
                    torch.cuda.synchronize()
                    return time.perf_counter() - t0

                public_loop(xs, 3)  # capture
                g_ms = 1e3 * public_loop(xs, 30) / 30
                D.GRAPH_SAMPLER = False
                public_loop(xs, 3)
                e_ms = 1e3 * public_loop(xs, 30) / 30
                D.GRAPH_SAMPLER = True
                x1 = xs[:1].contiguous()
                public_loop(x1, 3)  # capture at B = 1
                full_s = public_loop(x1, None) if full else None
                b1_ms = 1e3 * public_loop(x1, 50) / 50
                D.GRAPH_SAMPLER = False
                public_loop(x1, 3)
                b1_eager_ms = 1e3 * public_loop(x1, 50) / 50  # where the graph matters: one trajectory is host-bound in eager mode
                D.GRAPH_SAMPLER = True
                extra["sampling"]["public_p_sample_loop"] = {
                    "graph_ms_per_step": g_ms, "eager_ms_per_step": e_ms, "batch": Bs, "reverse_steps_timed": 30,
                    "measure_sample_time_B1_s": full_s, "B1_ms_per_reverse_step": b1_ms, "B1_eager_ms_per_reverse_step": b1_eager_ms,
                    "note": "GaussianDiffusion.p_sample_loop as the callers reach it; B = 1 full T = 1000 sample = the "
                            "reference's evaluate-runtime.py protocol" + ("" if full else " (full loop only with --steps >= 20)")}
            del sdiff
        leg_done("forward_and_sampling")
        if not multi:
            extra["cfg1"] = cfg1_leg(dev, with_cpu=not args.no_cpu_baseline)
            leg_done("cfg1_48x32x32")
            extra["real_grid"] = real_grid_leg(dev, B)
            extra["real_grid"]["per_voxel_vs_headline"] = (
                extra["real_grid"]["modes"]["bf16"]["ms_per_step"] / (REAL_GRID[0] * REAL_GRID[1] * REAL_GRID[2])
                / (out["ms_per_step"] / V)) if args.dtype == "bf16" else None
            leg_done("real_grid_194x50x50")
            extra["deterministic"] = deterministic_leg(diff, x, C, md, args.dtype, B, cell_idx)
            leg_done("deterministic_switch")
            extra["cfg5_attention"] = cfg5_attention_leg(dev)
            leg_done("cfg5_attention")
    if extra:
        out["extra"] = extra

    if not multi and not args.no_torch_baseline and not args.no_extra:
        del diff
        torch.cuda.empty_cache()
        # MIOpen's default find mode compiles and times every candidate solver of every conv shape on a cold cache: 614 s
        # of wall time for this leg on a fresh box (559 ms bf16 autocast / 4744 ms fp32 per step once it is done,
        # profiles/r10_bench_default_fullfind.json).  The default run asks for MIOpen's FAST mode (20 s; 587 / 6474 ms) and says
        # so; --torch-find full restores the exhaustive search.
        if args.torch_find == "fast":
            os.environ.setdefault("MIOPEN_FIND_MODE", "2")
        ref = {"miopen_find_mode": os.environ.get("MIOPEN_FIND_MODE", "default (exhaustive)"),
               "note": "exhaustive find on a cold cache takes ~10 min and gave 559 ms (bf16 autocast) / 4744 ms (fp32) per "
                       "step on this GPU (profiles/r10_bench_default_fullfind.json)"}
        for amp in (True, False):
            torch.cuda.reset_peak_memory_stats()
            try:
                ref["bf16_autocast" if amp else "fp32"] = torch_rocm_baseline(B, dev, amp, steps=2 if amp else 1)
            except Exception as e:  # noqa: BLE001 -- a baseline that cannot run is reported, not fatal
                ref["bf16_autocast" if amp else "fp32"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        out.setdefault("extra", {})["torch_rocm_reference"] = ref
        leg_done("torch_rocm_reference")
    if rank == 0 and wall:
        if "wall_s" in LIVE_TRAFFIC:
            wall["live_traffic_pmc_passes"] = LIVE_TRAFFIC["wall_s"]
        out.setdefault("extra", {})["wall_s"] = wall

    # ---- N > 1, LAST leg: the same step with forward + backward replayed from ONE captured graph under data parallelism: the
    # hooks' staging kernels are captured, every bucket's all-reduce is started by the host when the captured backward
    # passes the bucket's boundary (BucketedDataParallel.replay_launch) -- what un-binds the host where per-GPU batches are small.
    # It has never met a second GPU, and a rank that fails alone leaves the others inside a collective: so it runs after
    # everything else is measured, under a watchdog -- if it does not return in time, rank 0 prints the line it has (with a
    # note instead of this leg) and every rank leaves.
    if multi and not args.no_extra and fused_opt and "overlap" in extra:
        import threading

        from turbdiff_amd.training import GraphedTrainingStep

        def give_up():
            extra["overlap"]["captured_step"] = {"error": "did not finish within its 240-s watchdog; the line was printed without it"}
            if cpu is not None:
                out["cpu_baseline"] = cpu
            out["extra"] = extra
            if rank == 0:
                print(json.dumps(out), flush=True)
            os._exit(0)

        dog = threading.Timer(240.0, give_up)
        dog.daemon = True
        dog.start()
        task = _Task(diff, None)
        task.ddp = ddp
        opt_g = new_optimizer(diff, args.dtype, LOSS_ELEMENTS(B, cell_idx))
        task._opt = opt_g
        gs = GraphedTrainingStep(task)
        gbatch = SimpleNamespace(x=x, C=C, cell_idx=cell_idx)

        def graph_step():
            gs(gbatch)
            ddp.finish()
            opt_g.step()

        try:
            set_mode(diff, args.dtype)
            diff.zero_grad(set_to_none=True)
            for _ in range(3):
                graph_step()
            barrier()
            t0 = time.perf_counter()
            for _ in range(K):
                graph_step()
            enq = 1e3 * (time.perf_counter() - t0) / K
            barrier()
            t_graph = max_over_ranks(time.perf_counter() - t0) / K
            extra["overlap"]["captured_step"] = {"ms_per_step": 1e3 * t_graph, "host_enqueue_ms_per_step_rank0": enq,
                                                 "voxels_per_s": world * B * V / t_graph,
                                                 "note": "forward + backward as one hipGraph per rank; bucket all-reduces launched by "
                                                         "the host at the captured bucket boundaries; clip + RAdam eager"}
        except Exception as e:  # noqa: BLE001 -- reported, never fatal for the headline line
            extra["overlap"]["captured_step"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        dog.cancel()
        out["extra"] = extra

    if cpu is not None:
        out["cpu_baseline"] = cpu
    if rank == 0:
        print(json.dumps(out))
    if multi:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
