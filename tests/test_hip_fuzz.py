"""Randomised differential test of the three MFMA conv kernels against the vector-ALU kernels (which
are pinned to the oracle in test_hip_ops.py): random grids (ragged, thin-slab, permuted-brick and
tiny cases), channel counts (two inputs, half-filled tiles), batch sizes, with and without fused
GroupNorm statistics, residual addends and TDX_WS_CLEAN.  Seeds are fixed: the cases are reproducible."""

import random

import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu


def _cases(n, seed):
    rng = random.Random(seed)
    out = []
    for _ in range(n):
        kind = rng.choice(["small", "ragged", "thin", "flat"])
        if kind == "small":
            grid = (rng.randint(1, 9), rng.randint(1, 9), rng.randint(1, 9))
        elif kind == "ragged":
            grid = (rng.randint(5, 30), rng.randint(5, 20), rng.randint(5, 20))
        elif kind == "thin":  # >= 60000 voxels with 1-2 voxel remainders: the thin-slab path
            grid = (4 * rng.randint(10, 14) + rng.randint(1, 2), 8 * rng.randint(4, 5) + rng.randint(0, 2),
                    8 * rng.randint(4, 5) + rng.randint(0, 2))
        else:
            grid = (rng.randint(20, 60), rng.randint(3, 6), rng.randint(2, 4))
        c1 = rng.choice([16, 32, 32, 64, 64, 96])
        c2 = rng.choice([0, 0, 32, 64])
        co = rng.choice([32, 64, 96])
        out.append((grid, c1, c2, co, rng.randint(1, 3) if kind != "thin" else 1, rng.random() < 0.5, rng.randrange(1 << 30)))
    return out


@pytest.mark.parametrize("grid,C1,C2,Co,B,extras,seed", _cases(36, 1234) + _cases(28, 99))
def test_conv3_mfma_vs_direct_random(grid, C1, C2, Co, B, extras, seed):
    from turbdiff_amd import _lib as L, ops

    g = torch.Generator(device="cuda").manual_seed(seed)
    d = torch.device("cuda:0")
    X, Y, Z = grid
    Ci = C1 + C2
    rn = lambda *s: torch.randn(*s, device=d, generator=g)
    x1 = rn(B, X, Y, Z, C1).bfloat16()
    x2 = rn(B, X, Y, Z, C2).bfloat16() if C2 else None
    w = rn(Co, Ci, 3, 3, 3) * (2.0 / (27 * Ci)) ** 0.5
    bias = rn(Co)
    gy = rn(B, X, Y, Z, Co).bfloat16()
    st = L.stream()
    wf, wb = ops._packed_conv3(w, torch.bfloat16)

    def run(impl):
        y = torch.empty(B, X, Y, Z, Co, device=d, dtype=torch.bfloat16)
        stats = None
        if extras and Co % 8 == 0:
            stats = torch.empty(B, 8, 2, device=d)
            ws = torch.zeros(L.query("tdx_gn_workspace_bytes", B, Co), dtype=torch.uint8, device=d)
            L.call("tdx_conv3_fwd_gn", L.ptr(x1), C1, L.ptr(x2), C2, L.ptr(wf), L.ptr(bias), L.ptr(y), L.ptr(stats), 8, 1e-5,
                   L.ptr(ws), B, X, Y, Z, Co, L.BF16, impl | L.WS_CLEAN, st)
            assert int(ws.count_nonzero()) == 0
        else:
            L.call("tdx_conv3_fwd", L.ptr(x1), C1, L.ptr(x2), C2, L.ptr(wf), L.ptr(bias), L.ptr(y), B, X, Y, Z, Co, L.BF16, impl, st)
        gx1 = torch.empty_like(x1)
        gx2 = torch.empty_like(x2) if C2 else None
        dws = torch.empty(L.query("tdx_conv3_bwd_data_workspace_bytes", B, X, Y, Z, Ci, L.BF16, impl), dtype=torch.uint8, device=d)
        if extras:  # with the residual-path addends
            L.call("tdx_conv3_bwd_data_add", L.ptr(gy), L.ptr(wb), L.ptr(gx1), C1, L.ptr(gx2), C2, L.ptr(x1), L.ptr(x2), B, X, Y, Z,
                   Co, L.BF16, impl, L.ptr(dws), st)
        else:
            L.call("tdx_conv3_bwd_data", L.ptr(gy), L.ptr(wb), L.ptr(gx1), C1, L.ptr(gx2), C2, 0, B, X, Y, Z, Co, L.BF16, impl,
                   L.ptr(dws), st)
        gw, gb = torch.empty_like(w), torch.empty(Co, device=d)
        wws = torch.zeros(L.query("tdx_conv3_bwd_weight_workspace_bytes", Ci, Co, impl), dtype=torch.uint8, device=d)
        L.call("tdx_conv3_bwd_weight", L.ptr(x1), C1, L.ptr(x2), C2, L.ptr(gy), L.ptr(gw), L.ptr(gb), B, X, Y, Z, Co, L.BF16,
               impl | (L.WS_CLEAN if extras else 0), L.ptr(wws), st)
        return y, stats, gx1, gx2, gw, gb

    ref = run(L.CONV_DIRECT)
    got = run(L.CONV_AUTO)  # MFMA kernels wherever the shape is supported (the product's dispatch), else the same direct kernels
    names = ["y", "stats", "gx1", "gx2", "gw", "gb"]
    tols = {"y": 6e-3, "stats": 2e-3, "gx1": 8e-3, "gx2": 8e-3, "gw": 2e-3, "gb": 2e-3}
    for n, a, b in zip(names, got, ref):
        if a is None:
            assert b is None
            continue
        assert torch.isfinite(a.float()).all(), n
        assert rel_l2(a.float().cpu(), b.float().cpu()) < tols[n], (n, grid, C1, C2, Co, B)


# big grids with 32-wide output tiles: 8 x 8 x 8 bricks (split kernel) and thin remainder slabs of the padded
# data-gradient grid in one call
_BIG_CASES = [((98, 66, 50), 32, 0, 32, 4, True, 777), ((100, 64, 48), 32, 32, 32, 4, False, 778), ((192, 64, 48), 64, 0, 32, 1, True, 779)]


@pytest.mark.parametrize("impl_name", ["auto", "split"])
@pytest.mark.parametrize("grid,C1,C2,Co,B,extras,seed", _cases(14, 4321) + _BIG_CASES)
def test_conv3_fp32_mfma_and_split_vs_direct_random(grid, C1, C2, Co, B, extras, seed, impl_name, monkeypatch):
    """fp32 tensors: the IEEE-fp32 MFMA kernels (auto) and the split-precision kernels (split) against the
    vector-ALU kernels on random ragged / thin / tiny grids and channel mixes (two inputs, partly filled tiles)."""
    from turbdiff_amd import _lib as L, ops

    monkeypatch.setenv("TDX_CONV_IMPL", impl_name)
    g = torch.Generator(device="cuda").manual_seed(seed)
    d = torch.device("cuda:0")
    X, Y, Z = grid
    Ci = C1 + C2
    rn = lambda *s: torch.randn(*s, device=d, generator=g)
    x1, x2 = rn(B, X, Y, Z, C1), (rn(B, X, Y, Z, C2) if C2 else None)
    w = rn(Co, Ci, 3, 3, 3) * (2.0 / (27 * Ci)) ** 0.5
    bias, gy = rn(Co), rn(B, X, Y, Z, Co)
    st = L.stream()

    def run(impl):
        monkeypatch.setenv("TDX_CONV_IMPL", {L.CONV_DIRECT: "direct", L.CONV_AUTO: "auto", L.CONV_SPLIT: "split"}[impl])
        wf, wb = ops._packed_conv3(w, torch.float32)  # layout follows the mode (split images / kc = 8 / generic)
        y = torch.empty(B, X, Y, Z, Co, device=d)
        L.call("tdx_conv3_fwd", L.ptr(x1), C1, L.ptr(x2), C2, L.ptr(wf), L.ptr(bias), L.ptr(y), B, X, Y, Z, Co, L.F32, impl, st)
        gx1, gx2 = torch.empty_like(x1), (torch.empty_like(x2) if C2 else None)
        dws = torch.empty(L.query("tdx_conv3_bwd_data_workspace_bytes", B, X, Y, Z, Ci, L.F32, impl), dtype=torch.uint8, device=d)
        if extras:
            L.call("tdx_conv3_bwd_data_add", L.ptr(gy), L.ptr(wb), L.ptr(gx1), C1, L.ptr(gx2), C2, L.ptr(x1), L.ptr(x2), B, X, Y, Z,
                   Co, L.F32, impl, L.ptr(dws), st)
        else:
            L.call("tdx_conv3_bwd_data", L.ptr(gy), L.ptr(wb), L.ptr(gx1), C1, L.ptr(gx2), C2, 0, B, X, Y, Z, Co, L.F32, impl,
                   L.ptr(dws), st)
        gw, gb = torch.empty_like(w), torch.empty(Co, device=d)
        wws = torch.zeros(L.query("tdx_conv3_bwd_weight_workspace_bytes", Ci, Co, impl), dtype=torch.uint8, device=d)
        L.call("tdx_conv3_bwd_weight", L.ptr(x1), C1, L.ptr(x2), C2, L.ptr(gy), L.ptr(gw), L.ptr(gb), B, X, Y, Z, Co, L.F32,
               impl | (L.WS_CLEAN if extras else 0), L.ptr(wws), st)
        if extras:
            assert int(wws[: 27 * Ci * Co * 4 + Co * 4].count_nonzero()) == 0  # accumulators left clean
        return y, gx1, gx2, gw, gb

    ref = run(L.CONV_DIRECT)
    got = run(L.CONV_SPLIT if impl_name == "split" else L.CONV_AUTO)
    tol = 2e-5 if impl_name == "split" else 3e-6
    for n, a, b in zip(["y", "gx1", "gx2", "gw", "gb"], got, ref):
        if a is None:
            assert b is None
            continue
        assert torch.isfinite(a).all(), n
        assert rel_l2(a.cpu(), b.cpu()) < (1e-5 if n == "gb" else tol), (n, grid, C1, C2, Co, B)


def _small_cases(n, seed):
    rng = random.Random(seed)
    out = []
    for _ in range(n):
        grid = (rng.randint(1, 14), rng.randint(1, 11), rng.randint(1, 12))
        out.append((grid, rng.choice([8, 16, 24, 32, 64, 128]), rng.randint(1, 3), rng.choice([torch.float32, torch.bfloat16]),
                    rng.randrange(1 << 30)))
    return out


@pytest.mark.parametrize("grid,C,B,dtype,seed", _small_cases(24, 7))
def test_gn_film_silu_skip_resize_conv1_random(grid, C, B, dtype, seed):
    """GroupNorm + FiLM + SiLU + residual, skip_and_resize (both gradients merged in the resize adjoint) and the
    1x1 conv (two inputs + addend), forward and every gradient, against the same composition on stock torch
    ops in fp32: random grids / channel counts / groups / dtypes."""
    import torch.nn.functional as F

    from turbdiff_amd import ops

    rng = random.Random(seed)
    g = torch.Generator(device="cuda").manual_seed(seed)
    d = torch.device("cuda:0")
    X, Y, Z = grid
    G = rng.choice([k for k in (1, 2, 4, 8, C) if C % k == 0])
    Co = rng.choice([8, 32, 64])
    size = (rng.randint(1, 2 * X + 1), rng.randint(1, 2 * Y + 1), rng.randint(1, 2 * Z + 1))
    rn = lambda *s: torch.randn(*s, device=d, generator=g)
    leaf = lambda t: t.clone().requires_grad_()
    x, res = rn(B, X, Y, Z, C), rn(B, X, Y, Z, C)
    gamma, beta = 1 + 0.3 * rn(C), 0.2 * rn(C)
    scale, shift = 0.5 * rn(B, C), 0.5 * rn(B, C)
    w1, b1 = rn(Co, 2 * C) * (1.0 / (2 * C)) ** 0.5, rn(Co)
    add = rn(B, *size, Co)
    gy, gs = rn(B, *size, Co), rn(B, X, Y, Z, C)
    rd = lambda t: t.to(dtype).float()  # what the kernels see when activations are stored in `dtype`

    def torch_path(x, res, gamma, beta, scale, shift, w1, b1, add):
        ncv = lambda t: t.permute(0, 4, 1, 2, 3)
        nvc = lambda t: t.permute(0, 2, 3, 4, 1)
        n = F.group_norm(ncv(x), G, gamma, beta, 1e-5)
        n = n * (1 + scale)[:, :, None, None, None] + shift[:, :, None, None, None]
        h = nvc(F.silu(n)) + res
        r = nvc(F.interpolate(ncv(h), size=size, mode="trilinear", align_corners=True))
        y = torch.cat((r, r * 0.5), dim=-1) @ w1.t() + b1 + add
        return h, y

    def hip_path(x, res, gamma, beta, scale, shift, w1, b1, add):
        h = ops.gn_film_silu(x.to(dtype), gamma, beta, G, scale=scale, shift=shift, res=res.to(dtype))
        skip, r = ops.skip_and_resize(h, size)
        y = ops.conv1(r, w1, b1, x2=(r * 0.5).to(dtype), add=add.to(dtype))
        return skip, y

    ins = [x, res, gamma, beta, scale, shift, w1, b1, add]
    ta = [leaf(rd(t) if i in (0, 1, 8) else t) for i, t in enumerate(ins)]
    tb = [leaf(rd(t) if i in (0, 1, 8) else t) for i, t in enumerate(ins)]
    ha, ya = torch_path(*ta)
    hb, yb = hip_path(*tb)
    ((ya * gy).sum() + (ha * gs).sum()).backward()
    ((yb.float() * gy).sum() + (hb.float() * gs).sum()).backward()
    tol = 2e-4 if dtype == torch.float32 else 3e-2
    assert rel_l2(yb.float().cpu(), ya.cpu()) < tol and rel_l2(hb.float().cpu(), ha.cpu()) < tol
    for name, a, b in zip(["x", "res", "gamma", "beta", "scale", "shift", "w1", "b1", "add"], ta, tb):
        assert b.grad is not None and torch.isfinite(b.grad).all(), name
        if a.grad.norm() > 1e-6:
            assert rel_l2(b.grad.float().cpu(), a.grad.cpu()) < (2e-3 if dtype == torch.float32 else 6e-2), (name, grid, C, G, size)
