"""Pin oracle/grid_oracle.py (data ingress / egress) against tests/golden/grid_io.npz,
which tests/golden/make_golden_grid.py generated from the unmodified reference classes."""

import numpy as np
import pytest

from grid_cases import load_case
from oracle import grid_oracle as G


@pytest.fixture(scope="module", params=["A", "B"])
def case(request):
    return load_case(request.param)


def test_grid_embedding_bit_exact(case):
    x = G.grid_embedding(case.samples, case.variables, case.cell_idx, case.cell_counts, case.boundaries, case.fixed)
    assert x.dtype == np.float32 and x.shape == case.grid_embedding.shape
    assert np.array_equal(x, case.grid_embedding)


def test_normalizers_and_normalisation_bit_exact(case):
    assert len(case.modes) >= 7
    for mode, ref in case.modes.items():
        mean, std = G.normalizers(case.stats, case.variables, mode)
        assert np.array_equal(mean, ref.mean) and np.array_equal(std, ref.std), mode
        xn = G.normalize_grid(case.grid_embedding, mean, std)
        assert np.array_equal(xn, ref.normalized), mode
        assert np.array_equal(G.denormalize_grid(xn, mean, std), ref.denormalized), mode


def test_division_guard():
    c = load_case("B")
    _, std = G.normalizers(c.stats, c.variables, "std")
    assert std[-1] == 1.0  # k's std of 1e-9 is replaced (ofles.py:291)


def test_cell_types_and_embeddings(case):
    t = G.cell_types(case.cell_idx, case.cell_counts, case.boundaries)
    assert np.array_equal(t, case.cell_types)
    assert np.array_equal(G.cell_type_embedding(t, case.table), case.learned)
    assert np.array_equal(G.cell_type_onehot(t), case.onehot)
    g = G.cell_type_embedding_grad(t, case.grad_out)
    assert np.allclose(g, case.grad_table, rtol=1e-5, atol=1e-5)


def test_select_cells_channels_last(case):
    out = G.select_cells_channels_last(case.egress_x, case.cell_idx, case.variables)
    for name, _ in case.variables:
        assert np.array_equal(out[name], case.egress[name])


def test_round_trip_property():
    # size-independent: embedding then selecting the in-domain cells returns the samples wherever no
    # FIXED_VALUE boundary overwrote them (boundaries lie outside the cells in both cases)
    for tag in "AB":
        c = load_case(tag)
        x = G.grid_embedding(c.samples, c.variables, c.cell_idx, c.cell_counts, c.boundaries, c.fixed)
        back = G.select_cells_channels_last(x, c.cell_idx, c.variables)
        for name, _ in c.variables:
            assert np.array_equal(back[name], c.samples[name])
