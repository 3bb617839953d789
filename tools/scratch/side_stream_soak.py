"""Training-trajectory check of ops._WgradSide: 400 bf16 optimiser steps of the benchmark step (same seeds) twice with the
weight gradients on the launching stream and twice on the side stream.  The halo-shell kernel's bf16 atomics make single
runs differ (per-tensor gradient noise 2-3e-3 per step, tools/scratch/side_stream_race_check.py), so the comparison is
between the spread WITHIN a setting and the spread ACROSS settings, on a 20-step moving average of the loss."""
import os, sys
sys.path.insert(0, "tools"); sys.path.insert(0, "."); sys.path.insert(0, "generative-turbulence_amd")
import soak_train as S

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
runs = []
for sw in ("0", "0", "1", "1"):
    from turbdiff_amd import ops as _ops
    _ops.WGRAD_STREAM = sw == "1"  # (the environment variable is read once, at import)
    l, dt, mem = S.run("bf16", steps)
    runs.append((sw, l))
    marks = [i for i in (0, 100, 200, 300) if i < steps - 1] + [steps - 1]
    print(f"TDX_WGRAD_STREAM={sw}: {dt:.2f} ms/step, peak {mem:.1f} GB, loss at {'/'.join(map(str, marks))}: "
          + " ".join(f"{l[i]:.4f}" for i in marks), flush=True)
smooth = lambda l: [sum(l[i:i + 20]) / 20 for i in range(0, len(l) - 20)]
dev = lambda u, v: max(abs(a - b) / abs(a) for a, b in zip(smooth(u), smooth(v)))
print(f"largest relative deviation of the 20-step mean loss: launching stream run-to-run {dev(runs[0][1], runs[1][1]):.2e}; "
      f"side stream run-to-run {dev(runs[2][1], runs[3][1]):.2e}; across settings {dev(runs[0][1], runs[2][1]):.2e} / {dev(runs[1][1], runs[3][1]):.2e}")
