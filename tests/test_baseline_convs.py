"""Baseline conv variants on the HIP kernels: dilated 3x3x3 replicate convs (DilatedCNNBlock),
strided zero-padded convs + BatchNorm + LeakyReLU (tfnet conv()), transposed convs (tfnet deconv()) against the
golden vectors of the reference's own classes and against the CPU oracle on random shapes."""

import pytest
import torch

from conftest import rel_l2
from oracle import baselines_oracle as BO

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def nvc(x):
    return x.permute(0, 2, 3, 4, 1).contiguous()


def ncv(x):
    return x.permute(0, 4, 1, 2, 3).contiguous()


def _module(tag, g):
    from turbdiff_amd.models import baseline_convs as BC

    if tag.startswith("dil"):
        dil = [int(d) for d in g[f"{tag}/dilations"]]
        m = BC.DilatedCNNBlock(g[f"{tag}/x"].shape[1], dil)
    elif tag == "conv_s2":
        m = BC.conv(8, 16, kernel_size=3, stride=2, dropout_rate=0.0).train()
    elif tag == "conv_k5":
        m = BC.conv(8, 8, kernel_size=5, stride=2, dropout_rate=0.0).eval()
    else:
        m = BC.deconv(16, 8)
    m.load_state_dict(g.sub(f"{tag}/sd/"), strict=True)  # the reference module's own state_dict
    return m.to(dev())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("tag", ["dil", "dil8", "conv_s2", "conv_k5", "deconv"])
def test_baseline_layers_match_reference_vectors(golden, tag, dtype):
    g = golden("baselines")
    m = _module(tag, g)
    x = nvc(g[f"{tag}/x"]).to(dev()).to(dtype).requires_grad_()
    y = m(x)
    y.backward(nvc(g[f"{tag}/gy"]).to(dev()).to(dtype))
    # bf16 = bf16 activation storage between the (up to five) layers of a block; ReLU gates amplify a rounding that
    # flips a sign near zero, hence the loose gradient tolerance
    tol = 1e-5 if dtype == torch.float32 else 3e-2
    gtol = 1e-4 if dtype == torch.float32 else 0.1
    assert rel_l2(ncv(y.float().cpu()), g[f"{tag}/y"]) < tol
    assert rel_l2(ncv(x.grad.float().cpu()), g[f"{tag}/gx"]) < gtol
    for name, p in m.named_parameters():
        want = g[f"{tag}/grad/{name}"]
        if want.norm() < 1e-4:  # a conv bias in front of BatchNorm in training mode: mathematically zero (rounding noise)
            assert p.grad.norm() < 2e-3 * max(1.0, y.detach().float().norm().item())
        else:
            assert rel_l2(p.grad.cpu(), want) < gtol, name


@pytest.mark.parametrize("case", [
    # B, Cin, Cout, grid, k, stride, dilation, pad, mode
    (2, 8, 24, (7, 9, 5), 3, 1, 3, 3, "replicate"),
    (1, 48, 48, (12, 8, 10), 3, 1, 2, 2, "replicate"),   # DilResNet's hidden_dim
    (1, 16, 8, (9, 9, 8), 3, 2, 1, 1, "zeros"),
    (2, 8, 8, (6, 5, 7), 5, 1, 1, 2, "zeros"),
    (1, 8, 16, (8, 6, 6), 3, 1, 1, 0, "zeros"),           # "valid" convolution
])
def test_conv3d_general_vs_oracle(case):
    """ops.conv3d against F.conv3d on the CPU (the oracle's building block) incl. all gradients."""
    import torch.nn.functional as F

    from turbdiff_amd import ops

    B, Ci, Co, grid, k, s, d, p, mode = case
    gen = torch.Generator().manual_seed(1)
    x = torch.randn(B, Ci, *grid, generator=gen).requires_grad_()
    w = (torch.randn(Co, Ci, k, k, k, generator=gen) / (Ci * k**3) ** 0.5).requires_grad_()
    b = torch.randn(Co, generator=gen).requires_grad_()
    xp = F.pad(x, (p,) * 6, mode="replicate") if mode == "replicate" else x
    yr = F.conv3d(xp, w, b, stride=s, dilation=d, padding=0 if mode == "replicate" else p)
    gy = torch.randn(yr.shape, generator=gen)
    yr.backward(gy)
    xd = nvc(x.detach()).to(dev()).requires_grad_()
    wd, bd = w.detach().to(dev()).requires_grad_(), b.detach().to(dev()).requires_grad_()
    y = ops.conv3d(xd, wd, bd, stride=s, dilation=d, padding=p, padding_mode=mode)
    y.backward(nvc(gy).to(dev()))
    assert rel_l2(ncv(y.cpu()), yr) < 1e-5 and rel_l2(ncv(xd.grad.cpu()), x.grad) < 1e-5
    assert rel_l2(wd.grad.cpu(), w.grad) < 1e-4 and rel_l2(bd.grad.cpu(), b.grad) < 1e-4


def test_dilated_block_full_grid_properties():
    """DilatedCNNBlock at the benchmark grid (192 x 64 x 48, dim 48, dilations 1-2-4-8-4-2-1): too slow for the CPU
    oracle in a test, so size-independent properties -- a constant input gives a constant output equal to the same
    block on a tiny grid (replicate padding keeps constants constant), and the block is translation-covariant in the
    interior (receptive-field radius 22)."""
    from turbdiff_amd.models.baseline_convs import DilatedCNNBlock

    torch.manual_seed(0)
    blk = DilatedCNNBlock(48, [1, 2, 4, 8]).to(dev())
    with torch.no_grad():
        c = torch.randn(48, device=dev())
        big = blk(c.expand(1, 192, 64, 48, 48).contiguous())
        small = blk(c.expand(1, 4, 4, 4, 48).contiguous())
        assert (big - small[0, 0, 0, 0]).abs().max().item() < 1e-4
        x = torch.randn(1, 96, 64, 48, 48, device=dev())
        y = blk(x)
        ys = blk(torch.roll(x, shifts=5, dims=1))
        assert rel_l2(ys[:, 30:60], torch.roll(y, shifts=5, dims=1)[:, 30:60]) < 1e-5
