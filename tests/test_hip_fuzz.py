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
