#!/usr/bin/env python3
"""BASELINE config 4: T-step DDPM sampling of N trajectories sharded over the GPUs of one node,
one hipGraph-captured reverse step per replay (no collective inside the loop).

  python tools/sample_bench.py --trajectories 8 --timesteps 1000            # 1 GPU
  python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 tools/sample_bench.py --trajectories 64

Prints one JSON line (rank 0): whole-job samples/s = trajectories / max-over-ranks wall time."""
import argparse, json, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "generative-turbulence_amd"))
import torch
import bench
from turbdiff_amd import parallel
from turbdiff_amd.models.conditioning import Conditioning
from turbdiff_amd.sampling import GraphSampler

ap = argparse.ArgumentParser()
ap.add_argument("--trajectories", type=int, default=8)
ap.add_argument("--timesteps", type=int, default=1000)
ap.add_argument("--steps", type=int, default=0, help="time only this many reverse steps and extrapolate (0 = full loop)")
ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16", "f32", "f32s"])
ap.add_argument("--no-graph", action="store_true")
a = ap.parse_args()
if a.dtype == "f32s":  # fp32 tensors, split-precision convs
    import os
    os.environ["TDX_CONV_IMPL"] = "split"
rank, world, local = parallel.init_from_env("nccl")
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
diff = bench.build_model(dev, bench.MODE_DTYPE[a.dtype], timesteps=a.timesteps)
ids = list(parallel.shard_trajectories(a.trajectories, rank, world))
x, c, cell_idx = bench.synthetic_inputs(len(ids), dev)
C = {Conditioning.Type.CELL_TYPE: c}
s = GraphSampler(diff, x, C, cell_idx, seed=0, trajectory_ids=ids, use_graph=not a.no_graph)
s.run_steps(2); s.reset()          # warm-up incl. graph capture
torch.cuda.synchronize()
if world > 1: torch.distributed.barrier()
t0 = time.perf_counter()
n = a.steps if a.steps > 0 else a.timesteps
s.run_steps(n)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
if world > 1:
    t = torch.tensor([dt], device=dev, dtype=torch.float64)
    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX); dt = t.item()
if rank == 0:
    full = dt * a.timesteps / n
    print(json.dumps({"metric": "DDPM samples/sec (192x64x48x4)", "value": a.trajectories / full, "unit": "samples/s",
                      "n_gpus": world, "trajectories": a.trajectories, "per_gpu": len(ids), "timesteps": a.timesteps,
                      "timed_steps": n, "extrapolated": n != a.timesteps, "ms_per_reverse_step": 1e3 * dt / n,
                      "dtype": a.dtype, "hipgraph": not a.no_graph, "finite": bool(torch.isfinite(s.x_t).all())}))
if world > 1:
    torch.distributed.barrier(); torch.distributed.destroy_process_group()
