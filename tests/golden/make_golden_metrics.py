#!/usr/bin/env python3
"""Generate tests/golden/metrics.npz by running the *reference's* metric classes (build container only).

Imports the unmodified ``turbdiff.models.metrics`` (``interp3``, ``TurbulentKineticEnergySpectrum`` with the
reference's own Lebedev table, ``LogTKESpectrumL2Distance``) with the usual stand-ins for packages that are
not installed, feeds seeded inputs and stores inputs + outputs as numbers only.

    python tests/golden/make_golden_metrics.py

Cases: interp3 on an odd grid incl. out-of-range points (clamped corners, unclamped weights); the spectrum
with the reference's 5810-point rule on 14x12x10 and 16x16x16 (only the outputs are stored: the rule itself
is regenerated from scipy.integrate.lebedev_rule by the tests); the spectrum with a random 37-point rule
(stored) on 9x11x7; the log-spectrum distance with 8 Gauss-Legendre nodes.
"""
import sys
from pathlib import Path

import numpy as np
import torch

OUT = Path(__file__).resolve().parent
sys.path.insert(0, str(OUT))
from make_golden import REF, install_stubs, to_np  # noqa: E402


def main():
    install_stubs()
    sys.path.insert(0, str(REF))
    import turbdiff.models.metrics as M

    g = torch.Generator().manual_seed(424242)
    out = {}
    grid = torch.randn(2, 3, 7, 5, 6, generator=g)
    pts = torch.rand(40, 3, generator=g) * torch.tensor([8.0, 6.0, 7.0]) - 1.0  # some outside [0, n-1]
    out["interp3/grid"], out["interp3/points"] = to_np(grid), to_np(pts)
    out["interp3/out"] = to_np(M.interp3(grid, pts))

    spec = M.TurbulentKineticEnergySpectrum()  # the reference's numgrids.pickle, n = 5810
    out["lebedev/n"] = np.array(spec.n)
    out["lebedev/w_sum"] = to_np(spec.w.double().sum())
    for tag, shape, ks in (("a", (2, 3, 14, 12, 10), [1.0, 1.7, 2.5, 3.3, 4.0]), ("b", (3, 3, 16, 16, 16), [1.0, 2.0, 4.5, 7.0])):
        u = torch.randn(*shape, generator=g)
        k = torch.tensor(ks)
        out[f"spectrum/{tag}/u"], out[f"spectrum/{tag}/k"] = to_np(u), to_np(k)
        out[f"spectrum/{tag}/E"] = to_np(spec(u, k))

    # a random small rule, stored
    small = M.TurbulentKineticEnergySpectrum.__new__(M.TurbulentKineticEnergySpectrum)
    torch.nn.Module.__init__(small)
    p = torch.randn(37, 3, generator=g)
    p = p / p.norm(dim=1, keepdim=True)
    w = torch.rand(37, generator=g)
    w = w / w.sum()
    small.register_buffer("p", p)
    small.register_buffer("w", w)
    u = torch.randn(2, 3, 9, 11, 7, generator=g)
    k = torch.tensor([1.0, 1.5, 2.9])
    out["small/p"], out["small/w"], out["small/u"], out["small/k"] = to_np(p), to_np(w), to_np(u), to_np(k)
    out["small/E"] = to_np(small(u, k))

    dist = M.LogTKESpectrumL2Distance(small, n=8)
    ua, ub, um = torch.randn(3, 3, 9, 11, 7, generator=g), torch.randn(2, 3, 9, 11, 7, generator=g), 0.1 * torch.randn(3, 9, 11, 7, generator=g)
    D, la, lb, kk = dist(ua, ub, um)
    for name, t in (("u_a", ua), ("u_b", ub), ("u_mean", um), ("D", D), ("log_a", la), ("log_b", lb), ("k", kk),
                    ("nodes", dist.legendre_nodes), ("weights", dist.legendre_weights)):
        out[f"distance/{name}"] = to_np(t)
    np.savez_compressed(OUT / "metrics.npz", **out)
    print(f"wrote {OUT / 'metrics.npz'}: {(OUT / 'metrics.npz').stat().st_size / 1e3:.1f} kB")


if __name__ == "__main__":
    main()
