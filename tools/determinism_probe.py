#!/usr/bin/env python3
"""Is a training step bit-reproducible run to run?  Three identical steps (same weights, inputs, t, noise) of the benchmark model
at 96x32x24, B = 2, per mode, by default / with TDX_SHELL_DETERMINISTIC=1 / with TDX_DETERMINISTIC=1; lists the parameter
gradients that differ in any bit.  GPU box: python tools/determinism_probe.py [--grid 96 32 24] [--batch 2] [--reps 3]"""
import argparse, os, subprocess, sys
from pathlib import Path
from types import SimpleNamespace
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "generative-turbulence_amd"))

ap = argparse.ArgumentParser()
ap.add_argument("--grid", type=int, nargs=3, default=[96, 32, 24]); ap.add_argument("--batch", type=int, default=2)
ap.add_argument("--child", default=None); ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
if a.child is None:
    for mode in ("bf16", "fp16", "f32s"):
        for name, sw in (("default", {}), ("TDX_SHELL_DETERMINISTIC=1", {"TDX_SHELL_DETERMINISTIC": "1"}),
                         ("TDX_DETERMINISTIC=1", {"TDX_DETERMINISTIC": "1"})):
            env = {k: v for k, v in os.environ.items() if k not in ("TDX_SHELL_DETERMINISTIC", "TDX_DETERMINISTIC")}
            env.update(sw)
            out = subprocess.run([sys.executable, __file__, "--grid", *map(str, a.grid), "--batch", str(a.batch), "--reps", str(a.reps),
                                  "--child", mode], env=env, capture_output=True, text=True)
            lines = out.stdout.strip().splitlines()
            k = next((i for i, l in enumerate(lines) if l.startswith("loss identical")), None)
            print(f"{mode:5s} {name}: " + ("\n".join(lines[k:]) if k is not None else out.stderr[-600:]), flush=True)
    sys.exit(0)

import torch
import bench
from turbdiff_amd.models.conditioning import Conditioning

dev = torch.device("cuda:0")
diff = bench.build_model(dev)
bench.set_mode(diff, a.child)
x, c, idx = bench.synthetic_inputs(a.batch, dev, tuple(a.grid))
C, md = {Conditioning.Type.CELL_TYPE: c}, SimpleNamespace(cell_idx=idx)
t = torch.tensor(([3, 250] * a.batch)[: a.batch], device=dev)
noise = torch.randn(x.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
S = 2.0**12 if a.child == "fp16" else 1.0
runs = []
for rep in range(a.reps):
    diff.zero_grad(set_to_none=True)
    loss, _ = diff.p_losses(x, t, C, md, None, noise=noise)
    t0 = __import__("time").perf_counter()
    (loss * S).backward()
    torch.cuda.synchronize()
    bwd_ms = 1e3 * (__import__("time").perf_counter() - t0)
    runs.append((loss.item(), {n: p.grad.clone() for n, p in diff.model.named_parameters()}))
differ = sorted({n for r in runs[1:] for n in r[1] if not torch.equal(r[1][n], runs[0][1][n])})
worst = max(((runs[1][1][n] - runs[0][1][n]).norm() / runs[0][1][n].norm()).item() for n in differ) if differ else 0.0
print(f"loss identical {all(r[0] == runs[0][0] for r in runs)}; {len(differ)} of {len(runs[0][1])} parameter gradients differ between runs"
      f" (worst rel-L2 {worst:.1e})" + (": " + ", ".join(differ[:6]) + (" ..." if len(differ) > 6 else "") if differ else "")
      + f"; last backward {bwd_ms:.1f} ms")
if os.environ.get("TDX_PROBE_LIST") == "1":
    for n in differ:
        print("   ", n, f"{((runs[1][1][n] - runs[0][1][n]).norm() / runs[0][1][n].norm()).item():.1e}")
