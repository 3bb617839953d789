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


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("tag", ["dil", "dil8", "conv_s2", "conv_k5", "deconv"])
def test_baseline_layers_match_reference_vectors(golden, tag, dtype):
    g = golden("baselines")
    m = _module(tag, g)
    x = nvc(g[f"{tag}/x"]).to(dev()).to(dtype).requires_grad_()
    y = m(x)
    y.backward(nvc(g[f"{tag}/gy"]).to(dev()).to(dtype))
    # bf16 = bf16 activation storage between the (up to five) layers of a block; ReLU gates amplify a rounding that
    # flips a sign near zero, hence the loose gradient tolerance
    # fp16 tensors (round 6) take the vector-ALU kernels: 11-bit storage between the layers
    tol = {torch.float32: 1e-5, torch.bfloat16: 3e-2, torch.float16: 4e-3}[dtype]
    gtol = {torch.float32: 1e-4, torch.bfloat16: 0.1, torch.float16: 8e-2}[dtype]  # (measured: up to 5.6e-2 in fp16)
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


@pytest.mark.parametrize("case", [
    # B, Cin, Cout, grid, k, stride, dilation, pad, mode
    (2, 8, 24, (7, 9, 5), 3, 1, 3, 3, "replicate"),
    (1, 48, 48, (12, 8, 10), 3, 1, 2, 2, "replicate"),    # DilResNet's hidden_dim: 1.5 N tiles, 3 K steps
    (1, 48, 48, (20, 18, 17), 3, 1, 8, 8, "replicate"),   # dilation 8 on a grid barely wider than the stencil
    (1, 16, 8, (9, 9, 8), 3, 2, 1, 1, "zeros"),
    (2, 64, 128, (12, 10, 8), 3, 2, 1, 1, "zeros"),       # tfnet conv(): two N tiles per row, residue classes in the gradient
    (2, 8, 8, (6, 5, 7), 5, 1, 1, 2, "zeros"),
    (1, 8, 16, (8, 6, 6), 3, 1, 1, 0, "zeros"),           # "valid" convolution
    (1, 136, 72, (6, 5, 4), 3, 1, 1, 1, "zeros"),         # two weight stages, the second ending in a half K step
    (3, 24, 40, (5, 7, 6), 3, 3, 2, 2, "zeros"),          # stride 3 with dilation 2: 27 residue classes in the gradient
])
def test_conv3d_matrix_core_kernels_vs_oracle(case, monkeypatch):
    """bf16 tensors take the matrix-core kernels (tdx_convg_mfma.hip): forward and all three gradients against F.conv3d on
    the CPU in fp32 over the SAME bf16-rounded operands (the kernels round the fp32 weights to bf16, accumulate in fp32,
    store bf16 activations: 2^-9 per stored element), and against the vector-ALU kernels (TDX_CONVG_MFMA=0)."""
    import torch.nn.functional as F

    from turbdiff_amd import ops

    B, Ci, Co, grid, k, s, d, p, mode = case
    gen = torch.Generator().manual_seed(2)
    rb = lambda t: t.bfloat16().float()
    x = rb(torch.randn(B, Ci, *grid, generator=gen)).requires_grad_()
    w = rb(torch.randn(Co, Ci, k, k, k, generator=gen) / (Ci * k**3) ** 0.5).requires_grad_()
    b = torch.randn(Co, generator=gen).requires_grad_()
    xp = F.pad(x, (p,) * 6, mode="replicate") if mode == "replicate" else x
    yr = F.conv3d(xp, w, b, stride=s, dilation=d, padding=0 if mode == "replicate" else p)
    gy = rb(torch.randn(yr.shape, generator=gen))
    yr.backward(gy)

    def run():
        xd = nvc(x.detach()).to(dev()).bfloat16().requires_grad_()
        wd, bd = w.detach().to(dev()).requires_grad_(), b.detach().to(dev()).requires_grad_()
        y = ops.conv3d(xd, wd, bd, stride=s, dilation=d, padding=p, padding_mode=mode)
        y.backward(nvc(gy).to(dev()).bfloat16())
        return ncv(y.float().cpu()), ncv(xd.grad.float().cpu()), wd.grad.cpu(), bd.grad.cpu()

    y, gx, gw, gb = run()
    assert rel_l2(y, yr) < 4e-3 and rel_l2(gx, x.grad) < 4e-3
    assert rel_l2(gw, w.grad) < 1e-5 and rel_l2(gb, b.grad) < 1e-5
    monkeypatch.setenv("TDX_CONVG_MFMA", "0")
    y0, gx0, gw0, gb0 = run()
    assert rel_l2(y, y0) < 4e-3 and rel_l2(gx, gx0) < 4e-3 and rel_l2(gw, gw0) < 1e-5 and rel_l2(gb, gb0) < 1e-5


@pytest.mark.parametrize("case", [(2, 16, 8, (5, 6, 4), 4, 2, 1), (1, 128, 64, (6, 4, 5), 4, 2, 1), (2, 8, 24, (4, 5, 3), 3, 1, 1),
                                  (1, 40, 48, (5, 4, 6), 5, 3, 2)])
def test_conv_transpose3d_matrix_core_kernels_vs_oracle(case, monkeypatch):
    """tfnet deconv() (k 4, stride 2, padding 1) and other transposed convs in bf16 against F.conv_transpose3d in fp32 over
    the same bf16-rounded operands: the forward walks one residue class of output voxels per workgroup."""
    import torch.nn.functional as F

    from turbdiff_amd import ops

    B, Ci, Co, grid, k, s, p = case
    gen = torch.Generator().manual_seed(3)
    rb = lambda t: t.bfloat16().float()
    x = rb(torch.randn(B, Ci, *grid, generator=gen)).requires_grad_()
    w = rb(torch.randn(Ci, Co, k, k, k, generator=gen) / (Ci * k**3 / s**3) ** 0.5).requires_grad_()
    b = torch.randn(Co, generator=gen).requires_grad_()
    yr = F.conv_transpose3d(x, w, b, stride=s, padding=p)
    gy = rb(torch.randn(yr.shape, generator=gen))
    yr.backward(gy)

    def run():
        xd = nvc(x.detach()).to(dev()).bfloat16().requires_grad_()
        wd, bd = w.detach().to(dev()).requires_grad_(), b.detach().to(dev()).requires_grad_()
        y = ops.conv_transpose3d(xd, wd, bd, stride=s, padding=p)
        y.backward(nvc(gy).to(dev()).bfloat16())
        return ncv(y.float().cpu()), ncv(xd.grad.float().cpu()), wd.grad.cpu(), bd.grad.cpu()

    y, gx, gw, gb = run()
    assert rel_l2(y, yr) < 4e-3 and rel_l2(gx, x.grad) < 4e-3
    assert rel_l2(gw, w.grad) < 1e-5 and rel_l2(gb, b.grad) < 1e-4
    monkeypatch.setenv("TDX_CONVG_MFMA", "0")
    y0, gx0, gw0, _ = run()
    assert rel_l2(y, y0) < 4e-3 and rel_l2(gx, gx0) < 4e-3 and rel_l2(gw, gw0) < 1e-5


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
