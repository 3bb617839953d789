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
               FLOPs of all its launches / their HIP-event time, against the 2.5 PFLOP/s dense
               bf16 MFMA peak
  cpu_baseline the CPU oracle (a PyTorch-CPU port of the reference math, fp32) timed on the
               host cores on B = 1 of the same workload (N = 1, rank 0 only)
  extra        fwd-only and sampling (DDPM samples/s, extrapolated from a few reverse steps) legs
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
PEAK_BF16_TFLOPS = 2500.0  # dense, MI355X_MICROARCH.md
PEAK_F32_TFLOPS = 157.3


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


def build_model(device, dtype, timesteps=500):
    from turbdiff_amd.models.ddpm import DenoisingModel, GaussianDiffusion

    torch.manual_seed(0)
    net = DenoisingModel(in_features=4, out_features=4, c_local_features=4, c_global_features=0, timesteps=timesteps,
                         dim=32, u_net_levels=4, norm_type="group")
    net.set_compute_dtype(dtype)
    diff = GaussianDiffusion(net, timesteps=timesteps, beta_schedule="log-snr-linear", loss_type="l2", noise_bcs=True)
    return diff.to(device)


def cpu_baseline(budget_s=25.0):
    """fwd+bwd voxels/s of the CPU oracle on B=1 of the same workload."""
    from oracle import turbdiff_oracle as O

    torch.manual_seed(0)
    from turbdiff_amd.models.ddpm import DenoisingModel

    net = DenoisingModel(in_features=4, out_features=4, c_local_features=4, c_global_features=0, timesteps=500, dim=32,
                         u_net_levels=4, norm_type="group")
    sd = {k: v.clone().requires_grad_() for k, v in net.state_dict().items()}
    del net
    x, c, cell_idx = synthetic_inputs(1, "cpu")
    buf = O.schedule_buffers("log-snr-linear", 500)
    t = torch.tensor([250])
    noise = torch.randn(x.shape, generator=torch.Generator().manual_seed(1))
    times = []
    t_start = time.perf_counter()
    it = 0
    while True:
        t0 = time.perf_counter()
        loss, _ = O.p_losses(sd, buf, x, t, c, cell_idx, noise, timesteps=500, noise_bcs=True)
        torch.autograd.grad(loss, list(sd.values()))
        dt = time.perf_counter() - t0
        if it > 0:  # first iteration is warm-up
            times.append(dt)
        it += 1
        if len(times) >= 3 or (time.perf_counter() - t_start > budget_s and len(times) >= 1):
            break
    med = sorted(times)[len(times) // 2]
    V = GRID[0] * GRID[1] * GRID[2]
    return {"value": V / med, "unit": "voxels/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"B=1 fwd+bwd of the same 192x64x48 step, fp32, median of {len(times)} after 1 warm-up ({med:.2f} s each)"}


def torch_rocm_baseline(B, dev, amp, steps=2):
    """The reference's arithmetic path as it runs on this GPU through stock PyTorch-ROCm ops
    (MIOpen conv3d, ATen group_norm / interpolate / SDPA): the oracle's functional restatement on
    `cuda`, fp32 weights, optionally under bf16 autocast, same step (fwd + bwd + clip + RAdam).
    A reported baseline (--torch-baseline), never part of `value`."""
    from oracle import turbdiff_oracle as O
    from turbdiff_amd.models.ddpm import DenoisingModel

    torch.manual_seed(0)
    net = DenoisingModel(in_features=4, out_features=4, c_local_features=4, c_global_features=0, timesteps=500, dim=32,
                         u_net_levels=4, norm_type="group")
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
    V = GRID[0] * GRID[1] * GRID[2]
    res = {"ms_per_step": ms, "voxels_per_s": B * V / (ms * 1e-3), "batch": B, "precision": "bf16 autocast" if amp else "fp32",
           "peak_mem_GB": torch.cuda.max_memory_allocated() / 1e9}
    del sd, params, opt
    torch.cuda.empty_cache()
    return res


def measured_traffic(mode="bf16"):
    """HBM bytes per launch of the mode's forward conv kernel from the committed rocprofv3 PMC passes
    (profiles/*_traffic.json, produced by tools/collect_profiles.sh at B = 6): the newest file that holds
    launches of that kernel; (None, None) if there is none."""
    import re

    # forward instantiations = ZERO_PAD template argument false
    pat = {"bf16": r"conv3_mfma_kernel<\d, (true|false), false,",  # <NT, XT, ZERO_PAD, PERM, EXT>
           "f32": r"conv3_mfma_f32_kernel<\d, false,", "f32s": r"conv3_mfma_split_kernel<\d, false,"}[mode]
    for f in sorted((ROOT / "profiles").glob("*_traffic.json"), reverse=True):  # rNN tags: newest round first
        data = json.loads(f.read_text())
        num = den = 0.0
        for name, k in data["kernels"].items():
            if re.search(pat, name):
                n = k["launches_sampled"]
                num += n * (k["read_bytes_per_launch"] + k["write_bytes_per_launch"])
                den += n
        if den:
            return num / den, f"profiles/{f.name}"
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=6, help="per-GPU batch (reference: 6, config/model/diffusion.yaml:3)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32", "f32s"],
                    help="bf16: bf16 storage + bf16 MFMA (BASELINE configs[1]); f32: fp32 storage, IEEE fp32 MFMA convs "
                         "(the 1e-4 parity mode); f32s: fp32 storage, split-precision convs (bf16 hi + lo, 3 MFMAs per "
                         "product: 1.2e-5 rel-L2 at full size, still inside the 1e-4 gate)")
    ap.add_argument("--sample-steps", type=int, default=6, help="reverse steps timed for the sampling leg (0 = skip)")
    ap.add_argument("--sample-batch", type=int, default=8)
    ap.add_argument("--optimizer", default="fused", choices=["fused", "torch"],
                    help="fused: ClipRAdam (clip + RAdam in 3 launches); torch: clip_grad_norm_ + torch.optim.RAdam")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true")
    ap.add_argument("--torch-baseline", action="store_true",
                    help="also time the same step through stock PyTorch-ROCm ops (MIOpen); adds extra.torch_rocm_reference")
    args = ap.parse_args()

    from turbdiff_amd import _lib, parallel

    # TDX_BENCH_BACKEND=gloo + TDX_BENCH_ONE_DEVICE=1: all ranks on cuda:0 with gloo collectives -- only for
    # exercising the N > 1 code path on a one-GPU box (tests/test_bench_multirank.py); never a measurement
    rank, world, local = parallel.init_from_env(os.environ.get("TDX_BENCH_BACKEND", "nccl"))
    if os.environ.get("TDX_BENCH_ONE_DEVICE") == "1":
        local = 0
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    if args.dtype == "f32s":
        os.environ["TDX_CONV_IMPL"] = "split"
    B, K, Wm = args.batch, args.steps, args.warmup
    V = GRID[0] * GRID[1] * GRID[2]

    diff = build_model(dev, dtype)
    ddp = parallel.BucketedDataParallel(diff)
    fused_opt = args.optimizer == "fused"
    if fused_opt:  # clip 0.1 + RAdam in three launches (turbdiff_amd/optim.py)
        from turbdiff_amd.optim import ClipRAdam

        opt = ClipRAdam(diff.parameters(), lr=1e-4, max_norm=0.1)
    else:
        opt = torch.optim.RAdam(diff.parameters(), lr=1e-4)
    x, c_local, cell_idx = synthetic_inputs(B, dev)
    from turbdiff_amd.models.conditioning import Conditioning

    C = {Conditioning.Type.CELL_TYPE: c_local}
    md = SimpleNamespace(cell_idx=cell_idx)

    def train_step():
        loss, _ = diff(x, C, md, None)
        loss.backward()
        ddp.finish()
        if not fused_opt:
            torch.nn.utils.clip_grad_norm_(diff.parameters(), 0.1)
        opt.step()
        opt.zero_grad(set_to_none=True)
        return loss

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(Wm):
        train_step()
    timer = _lib.KernelTimer({"tdx_conv3_fwd", "tdx_conv3_fwd_gn", "tdx_conv3_bwd_data", "tdx_conv3_bwd_data_add",
                               "tdx_conv3_bwd_weight"})
    _lib.TIMER = timer
    barrier()
    t0 = time.perf_counter()
    for _ in range(K):
        loss = train_step()
    barrier()
    elapsed = time.perf_counter() - t0
    _lib.TIMER = None
    kern = timer.summary()
    # tdx_conv3_fwd_gn = the same conv kernel with the GroupNorm statistics in its epilogue
    # (+ a memset and an 8 us finalize kernel inside the bracket): count it as a forward launch
    for alias, name in (("tdx_conv3_fwd_gn", "tdx_conv3_fwd"), ("tdx_conv3_bwd_data_add", "tdx_conv3_bwd_data")):
        if alias in kern:
            f = kern.setdefault(name, {"launches": 0, "ms": 0.0, "work": 0.0})
            g = kern.pop(alias)
            for k in f:
                f[k] += g[k]
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        elapsed = tt.item()
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
                   "optimizer": "ClipRAdam (fused clip 0.1 + RAdam)" if fused_opt else "clip_grad_norm_ + torch.optim.RAdam"},
        "loss": float(loss),
    }
    if rank == 0:
        # f32s issues three bf16 MFMAs per algorithmic product: its matrix-core ceiling is a third of the bf16 peak
        peak = {"bf16": PEAK_BF16_TFLOPS, "f32": PEAK_F32_TFLOPS, "f32s": PEAK_BF16_TFLOPS / 3}[args.dtype]
        kf = kern.get("tdx_conv3_fwd")
        if kf and kf["ms"] > 0:
            ach = kf["work"] / (kf["ms"] * 1e-3) / 1e12
            traffic, tsrc = measured_traffic(args.dtype)
            out["roofline"] = {"bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                               "traffic": traffic, "traffic_unit": "HBM bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE)",
                               "traffic_source": tsrc,
                               "algorithmic_bytes_per_launch": (1417.6e6 if dtype == torch.bfloat16 else 2835.2e6) * B / 22,
                               "kernel": {"bf16": "conv3_mfma_kernel", "f32": "conv3_mfma_f32_kernel",
                                          "f32s": "conv3_mfma_split_kernel"}[args.dtype] + " (all tdx_conv3_fwd launches)",
                               "launches": kf["launches"], "avg_launch_ms": kf["ms"] / kf["launches"],
                               "conv_bandwidth_roofline_frac": (1417.6e6 * B * K / (kf["ms"] * 1e-3)) / 8e12
                               if dtype == torch.bfloat16 else (2835.2e6 * B * K / (kf["ms"] * 1e-3)) / 8e12}
        out["kernels"] = {k: {"launches": v["launches"], "ms_per_step": v["ms"] / K,
                              "tflops": v["work"] / max(v["ms"], 1e-9) / 1e9} for k, v in kern.items()}

    if not args.no_extra and world == 1:
        extra = {}
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
            from turbdiff_amd.sampling import GraphSampler

            Bs = args.sample_batch
            xs, _, _ = synthetic_inputs(Bs, dev)
            sampler = GraphSampler(diff, xs, C, cell_idx, seed=0)
            sampler.run_steps(2)  # warm-up (includes graph capture)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            sampler.run_steps(args.sample_steps)
            torch.cuda.synchronize()
            per_step = (time.perf_counter() - t0) / args.sample_steps
            extra["sample_ms_per_step"] = 1e3 * per_step
            extra["sample_batch"] = Bs
            extra["ddpm_samples_per_s_T500_extrapolated"] = Bs / (per_step * 500)
            extra["ddpm_samples_per_s_T1000_extrapolated"] = Bs / (per_step * 1000)
        out["extra"] = extra

    if args.torch_baseline and world == 1:
        del diff, ddp, opt
        torch.cuda.empty_cache()
        ref = {}
        for amp in (True, False):
            torch.cuda.reset_peak_memory_stats()
            try:
                ref["bf16_autocast" if amp else "fp32"] = torch_rocm_baseline(B, dev, amp)
            except Exception as e:  # noqa: BLE001 -- a baseline that cannot run is reported, not fatal
                ref["bf16_autocast" if amp else "fp32"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        out.setdefault("extra", {})["torch_rocm_reference"] = ref

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline()
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
