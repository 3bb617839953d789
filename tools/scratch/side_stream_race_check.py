"""Race check of ops._WgradSide: the gradients of ONE fixed training batch (B = 6, 192x64x48, bf16) computed with the weight
gradients on the launching stream (reference) and then `reps` times on the side stream; prints the largest per-tensor
rel-L2 deviation seen in any repetition (what is left must be the bf16 atomics of the halo-shell kernel: <= 1e-2)."""
import os, sys
sys.path.insert(0, "."); sys.path.insert(0, "generative-turbulence_amd")
from types import SimpleNamespace
import torch
import bench
from turbdiff_amd.models.conditioning import Conditioning

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = torch.device("cuda:0")
torch.manual_seed(0)
diff = bench.build_model(dev, torch.bfloat16)
x, c, idx = bench.synthetic_inputs(6, dev)
C = {Conditioning.Type.CELL_TYPE: c}; md = SimpleNamespace(cell_idx=idx)
t = torch.randint(0, 500, (6,), device=dev)
noise = torch.randn_like(x)


def grads(stream):
    from turbdiff_amd import ops as _ops
    _ops.WGRAD_STREAM = stream == "1"  # (the environment variable is read once, at import)
    diff.zero_grad(set_to_none=True)
    loss, _ = diff.p_losses(x, t, C, md, None, noise=noise)
    loss.backward()
    torch.cuda.synchronize()
    return {n: p.grad.detach().float().clone() for n, p in diff.model.named_parameters() if p.grad is not None}, loss.item()


ref, l0 = grads("0")
ref2, _ = grads("0")
rel = lambda a, b: ((a - b).norm() / b.norm().clamp_min(1e-30)).item()
base = max(rel(ref2[n], ref[n]) for n in ref)
print(f"launching stream, run-to-run: largest per-tensor rel-L2 {base:.2e} (halo-shell atomics)", flush=True)
worst, where = 0.0, None
for r in range(reps):
    g, l = grads("1")
    for n in ref:
        d = rel(g[n], ref[n])
        if d > worst:
            worst, where = d, (r, n)
    if not all(torch.isfinite(v).all() for v in g.values()):
        print("non-finite gradient in repetition", r); break
print(f"side stream vs launching stream over {reps} repetitions: largest per-tensor rel-L2 {worst:.2e} at {where}; loss {l0:.6f} / {l:.6f}")
