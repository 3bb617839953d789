#!/usr/bin/env python3
"""TKE spectrum at the reference's region size: 46^3, 5810 Lebedev nodes, 64 radii, B samples --
HIP path vs the reference's op chain through stock PyTorch-ROCm on the same GPU.  GPU box: python tools/metrics_bench.py"""
import argparse, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "generative-turbulence_amd"))
import torch
from turbdiff_amd.models import metrics as M


def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=8); a = ap.parse_args()
    spec = M.TurbulentKineticEnergySpectrum().cuda()
    u = torch.randn(a.batch, 3, 46, 46, 46, device="cuda"); k = torch.linspace(1.0, 22.0, 64, device="cuda")

    def ref():  # metrics.py:289-316 verbatim in torch ops
        tke = 0.5 * (u ** 2).sum(dim=-4)
        f = torch.fft.fftshift(torch.fft.fftn(tke, dim=(-3, -2, -1)), dim=(-3, -2, -1))
        q = k[:, None, None] * spec.p + k.new_tensor([23.0, 23.0, 23.0])
        return torch.matmul(M.interp3((f.abs() ** 2).log(), q).exp().float(), spec.w) * (4 * torch.pi * k ** 2)

    err = ((spec(u, k) - ref()).abs() / ref().abs()).max().item()
    t, tr = timeit(lambda: spec(u, k)), timeit(ref)
    q = a.batch * 64 * 5810
    print(f"B = {a.batch}, 46^3, 5810 nodes x 64 radii = {q/1e6:.2f} M queries: HIP {t*1e3:.0f} us ({q/t/1e6:.1f} G queries/s), "
          f"torch-ROCm op chain {tr*1e3:.0f} us, {tr/t:.1f}x; max rel diff {err:.1e}")


if __name__ == "__main__":
    main()
