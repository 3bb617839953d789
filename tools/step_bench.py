#!/usr/bin/env python3
"""One training step (fwd + bwd + clip + RAdam) of the BASELINE configs[1] U-Net on an arbitrary grid, for profiling:
    python tools/step_bench.py --grid 194 50 50 --mode bf16 --batch 6 --steps 5
prints ms per step; under `rocprofv3 --kernel-trace --stats` the per-kernel table of exactly these steps."""
import argparse, sys, time
from pathlib import Path
from types import SimpleNamespace

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "generative-turbulence_amd"))
import torch
import bench
from turbdiff_amd.models.conditioning import Conditioning

ap = argparse.ArgumentParser()
ap.add_argument("--grid", type=int, nargs=3, default=[194, 50, 50])
ap.add_argument("--mode", default="bf16")
ap.add_argument("--batch", type=int, default=6)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--warmup", type=int, default=2)
a = ap.parse_args()
dev = torch.device("cuda:0")
diff = bench.build_model(dev)
x, c, idx = bench.synthetic_inputs(a.batch, dev, tuple(a.grid))
C, md = {Conditioning.Type.CELL_TYPE: c}, SimpleNamespace(cell_idx=idx)
ms = bench.timed_train_steps(diff, x, C, md, a.mode, a.steps, a.warmup)
v = a.grid[0] * a.grid[1] * a.grid[2]
print(f"grid {a.grid} B {a.batch} {a.mode}: {ms:.3f} ms/step = {a.batch * v / ms / 1e3:.1f} M voxels/s")
