"""Does any kernel of the training step read memory nobody wrote?  Every torch.empty / empty_like of the step is filled with NaN
(byte workspaces with 0xFF) before use; an uninitialised read shows up as a non-finite loss or gradient.  Two eager steps of the
two-level golden model (f32) or of the benchmark model at 96x32x24 in a given mode.  Round 6: clean in every mode.
GPU box: python tools/uninit_read_probe.py [f32 | full bf16 | full fp16 | full f32s]"""
import os, sys
from pathlib import Path
from types import SimpleNamespace
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tests")); sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "generative-turbulence_amd"))
import torch
_empty, _empty_like = torch.empty, torch.empty_like
def nan_fill(t):
    if t.is_floating_point() and t.is_cuda: t.fill_(float("nan"))
    elif t.is_cuda and t.dtype == torch.uint8: t.fill_(0xFF)  # byte workspaces: all-ones bytes = NaN patterns in any float view
    return t
torch.empty = lambda *a, **k: nan_fill(_empty(*a, **k))
torch.empty_like = lambda *a, **k: nan_fill(_empty_like(*a, **k))
import test_parallel_gpu as T
dev = torch.device("cuda:0")
mode = sys.argv[1] if len(sys.argv) > 1 else "f32"
if mode == "full":
    import bench
    from turbdiff_amd.models.conditioning import Conditioning
    diff = bench.build_model(dev); bench.set_mode(diff, sys.argv[2])
    x, c, idx = bench.synthetic_inputs(2, dev, (96, 32, 24))
    C, md = {Conditioning.Type.CELL_TYPE: c}, SimpleNamespace(cell_idx=idx)
    t = torch.tensor([3, 250], device=dev); noise = torch.randn(x.shape, device=dev)
else:
    diff, x, C, md, t, noise = T._build(dev)
for rep in range(2):
    diff.zero_grad(set_to_none=True)
    loss, _ = diff.p_losses(x, t, C, md, None, noise=noise)
    loss.backward()
    torch.cuda.synchronize()
    bad = [n for n, p in diff.model.named_parameters() if p.grad is None or not torch.isfinite(p.grad).all()]
    print("rep", rep, "loss", loss.item(), "non-finite grads:", bad[:10], len(bad), flush=True)
