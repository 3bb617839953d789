"""On-device sample metrics: the oracle against the golden vectors generated from the
unmodified reference (tests/golden/make_golden_metrics.py), and the HIP path against both."""

import numpy as np
import pytest
import torch

from oracle import metrics_oracle as MO
from turbdiff_amd.models import metrics as M


@pytest.fixture(scope="module")
def g(golden):
    return golden("metrics")


# ------------------------------------------------------------------------------------ CPU
def test_interp3_oracle_and_host_bit_exact(g):
    want = g["interp3/out"]
    assert torch.equal(MO.interp3(g["interp3/grid"], g["interp3/points"]), want)
    assert torch.equal(M.interp3(g["interp3/grid"], g["interp3/points"]), want)


def test_scipy_lebedev_rule_is_the_reference_rule(g):
    p, w = M.lebedev_rule(int(g["lebedev/n"]))
    assert p.shape == (5810, 3) and abs(w.double().sum().item() - 1) < 1e-6
    assert torch.allclose(p.norm(dim=1), torch.ones(5810), atol=1e-6)
    # the oracle with scipy's rule reproduces the reference's spectra (computed with numgrids.pickle)
    for tag in "ab":
        E = MO.tke_spectrum(g[f"spectrum/{tag}/u"], g[f"spectrum/{tag}/k"], p, w)
        assert torch.allclose(E, g[f"spectrum/{tag}/E"], rtol=2e-5, atol=0)
    with pytest.raises(RuntimeError):
        M.lebedev_rule(5811)


def test_spectrum_and_distance_oracle(g):
    E = MO.tke_spectrum(g["small/u"], g["small/k"], g["small/p"], g["small/w"])
    assert torch.allclose(E, g["small/E"], rtol=1e-6, atol=0)
    D, la, lb, k = MO.log_tke_distance(g["distance/u_a"], g["distance/u_b"], g["distance/u_mean"], g["small/p"], g["small/w"],
                                       g["distance/nodes"], g["distance/weights"])
    assert torch.equal(k, g["distance/k"])
    assert torch.allclose(la, g["distance/log_a"], rtol=1e-6, atol=1e-6) and torch.allclose(lb, g["distance/log_b"], rtol=1e-6, atol=1e-6)
    assert torch.allclose(D, g["distance/D"], rtol=1e-5, atol=1e-6)


def test_module_surface():
    d = M.LogTKESpectrumL2Distance(M.TurbulentKineticEnergySpectrum(n=50), n=8)
    assert sorted(d.state_dict()) == ["legendre_nodes", "legendre_weights", "tke_spectrum.p", "tke_spectrum.w"]
    assert d.tke_spectrum.p.shape == (50, 3)
    with pytest.raises(RuntimeError, match="device tensors"):
        d.tke_spectrum(torch.zeros(1, 3, 4, 4, 4), torch.ones(2))


# ------------------------------------------------------------------------------------ GPU
def dev():
    return torch.device("cuda:0")


TOL = 2e-5  # fp32 rel: device log / exp / FFT vs the CPU's; sums over 37-5810 positive terms


@pytest.mark.gpu
def test_spectrum_small_rule_vs_golden(g):
    spec = M.TurbulentKineticEnergySpectrum(n=50).to(dev())
    spec.p, spec.w = g["small/p"].to(dev()), g["small/w"].to(dev())
    E = spec(g["small/u"].to(dev()), g["small/k"].to(dev()))
    assert torch.allclose(E.cpu(), g["small/E"], rtol=TOL, atol=0)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["a", "b"])
def test_spectrum_lebedev_5810_vs_golden(g, tag):
    spec = M.TurbulentKineticEnergySpectrum().to(dev())
    E = spec(g[f"spectrum/{tag}/u"].to(dev()), g[f"spectrum/{tag}/k"].to(dev()))
    assert E.shape == g[f"spectrum/{tag}/E"].shape
    assert torch.allclose(E.cpu(), g[f"spectrum/{tag}/E"], rtol=TOL, atol=0)


@pytest.mark.gpu
def test_distance_vs_golden(g):
    spec = M.TurbulentKineticEnergySpectrum(n=50)
    dist = M.LogTKESpectrumL2Distance(spec, n=8).to(dev())
    spec.p, spec.w = g["small/p"].to(dev()), g["small/w"].to(dev())
    D, la, lb, k = dist(g["distance/u_a"].to(dev()), g["distance/u_b"].to(dev()), g["distance/u_mean"].to(dev()))
    assert torch.allclose(k.cpu(), g["distance/k"], rtol=1e-6)
    assert torch.allclose(la.cpu(), g["distance/log_a"], rtol=1e-5, atol=1e-5)
    assert torch.allclose(D.cpu(), g["distance/D"], rtol=1e-4, atol=1e-5)


@pytest.mark.gpu
def test_full_region_size_properties():
    """46^3 regions as WassersteinTKE cuts them (metrics.py:425-450), 64 radii, 5810 nodes."""
    spec = M.TurbulentKineticEnergySpectrum().to(dev())
    gen = torch.Generator().manual_seed(3)
    u = torch.randn(4, 3, 46, 46, 46, generator=gen).to(dev())
    k = torch.linspace(1.0, 22.0, 64, device=dev())
    E = spec(u, k)
    assert E.shape == (4, 64) and torch.isfinite(E).all() and (E > 0).all()
    # against the reference's op chain run through torch on the device
    tke = 0.5 * (u ** 2).sum(dim=-4)
    f = torch.fft.fftshift(torch.fft.fftn(tke, dim=(-3, -2, -1)), dim=(-3, -2, -1))
    q = k[:, None, None] * spec.p + k.new_tensor([23.0, 23.0, 23.0])
    ref = torch.matmul(M.interp3((f.abs() ** 2).log(), q).exp().float(), spec.w) * (4 * torch.pi * k ** 2)
    assert torch.allclose(E, ref, rtol=5e-5, atol=0)
    # E(k) scales with the fourth power of the velocity amplitude (|FFT(tke)|^2, tke ~ u^2)
    assert torch.allclose(spec(2 * u, k), 16 * E, rtol=1e-4)
    # the order of the quadrature nodes does not matter (only the rounding of the sum does)
    perm = torch.randperm(spec.p.shape[0], generator=gen).to(dev())
    spec2 = M.TurbulentKineticEnergySpectrum().to(dev())
    spec2.p, spec2.w = spec.p[perm].contiguous(), spec.w[perm].contiguous()
    assert torch.allclose(spec2(u, k), E, rtol=1e-5, atol=0)
