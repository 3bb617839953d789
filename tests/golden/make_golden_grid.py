#!/usr/bin/env python3
"""Generate tests/golden/grid_io.npz by running the *reference's* data ingress / egress code.

Runs only in the build container (reference mounted read-only at /root/reference).  It imports the
reference's unmodified ``turbdiff.data.ofles`` (``OpenFOAMData.grid_embedding``, ``OpenFOAMStats
.normalizers``), ``turbdiff.models.normalization.Normalization``, ``turbdiff.models
.cell_type_embeddings`` and ``turbdiff.models.utils.select_cells`` -- with the same empty stand-ins
for h5py / lightning / cachetools as ``make_golden.py`` -- feeds them two small synthetic geometries
and stores inputs and outputs as numbers only.

    python tests/golden/make_golden_grid.py

Cases:
  A  11 x 8 x 7 padded grid, variables (u, p), an obstacle, walls / inlets / outlets boundaries that
     overlap on edges (so the write order of ofles.py:235-238 matters), FIXED_VALUE u on walls and
     inlets, FIXED_VALUE p on outlets, unsorted cell_idx
  B  9 x 3 x 6 two-dimensional grid with an ``empties`` boundary, variables (p, u, k)
and for each: grid_embedding, normalizers for every mode, normalize / denormalize, cell types, the
learned and one-hot embeddings (+ the table gradient for a seeded upstream gradient), and the
select_cells + channels-last split that SampleStore.add_samples writes (metrics.py:52-58).
"""

import sys
from pathlib import Path

import numpy as np
import torch

OUT = Path(__file__).resolve().parent
sys.path.insert(0, str(OUT))
from make_golden import REF, install_stubs, to_np  # noqa: E402

MODES = ["u:norm-max;p:abs-max", "mean-std", "std", "abs-max", "norm", "norm-std", "norm-max",
         "u:mean-std;p:std;k:abs-max"]


def geometry_a(rng):
    n = (11, 8, 7)
    inside = np.zeros(n, dtype=bool)
    inside[1:-1, 1:-1, 1:-1] = True
    inside[3:6, 1:4, 2:5] = False  # obstacle
    cell_idx = np.flatnonzero(inside.reshape(-1))
    cell_idx = cell_idx[rng.permutation(len(cell_idx))]  # the reference does not need it sorted
    g = np.arange(np.prod(n)).reshape(n)
    walls = np.concatenate([g[:, 0, :].ravel(), g[:, -1, :].ravel(), g[:, :, 0].ravel(), g[:, :, -1].ravel(),
                            g[3:6, 1:4, 2:5].ravel()])
    inlets = g[0, :, :].ravel()      # overlaps the walls on the rim
    outlets = g[-1, :, :].ravel()
    boundaries = {"walls": walls, "inlets": inlets, "outlets": outlets}
    fixed = {"u": {"walls": np.zeros(3, np.float32), "inlets": np.array([1.3, 0.1, -0.2], np.float32)},
             "p": {"outlets": np.array(0.7, np.float32)}}
    zero_grad = {"u": ["outlets"], "p": ["walls", "inlets"]}
    return n, cell_idx, boundaries, fixed, zero_grad, (("u", 3), ("p", 1))


def geometry_b(rng):
    n = (9, 3, 6)
    inside = np.zeros(n, dtype=bool)
    inside[1:-1, 1:-1, 1:-1] = True
    cell_idx = np.flatnonzero(inside.reshape(-1))
    g = np.arange(np.prod(n)).reshape(n)
    boundaries = {"empties": np.concatenate([g[:, 0, :].ravel(), g[:, -1, :].ravel()]),
                  "walls": np.concatenate([g[:, :, 0].ravel(), g[:, :, -1].ravel()]),
                  "inlets": g[0, 1:-1, 1:-1].ravel(), "outlets": g[-1, 1:-1, 1:-1].ravel()}
    fixed = {"p": {"outlets": np.array(0.0, np.float32)},
             "u": {"inlets": np.array([2.0, 0.0, 0.0], np.float32), "walls": np.zeros(3, np.float32)},
             "k": {"walls": np.array(1e-3, np.float32), "inlets": np.array(0.05, np.float32)}}
    zero_grad = {"p": ["walls", "inlets"], "u": ["outlets"], "k": ["outlets"]}
    return n, cell_idx, boundaries, fixed, zero_grad, (("p", 1), ("u", 3), ("k", 1))


def main():
    install_stubs()
    sys.path.insert(0, str(REF))
    torch.use_deterministic_algorithms(True)
    from turbdiff.data.ofles import BoundaryCondition as BC
    from turbdiff.data.ofles import OpenFOAMData, OpenFOAMMetadata, OpenFOAMStats
    from turbdiff.data.ofles import Variable as V
    from turbdiff.models.cell_type_embeddings import CellTypeEmbedding
    from turbdiff.models.normalization import Normalization
    from turbdiff.models.utils import select_cells

    out = {}
    rng = np.random.default_rng(20240917)
    for tag, geo in (("A", geometry_a), ("B", geometry_b)):
        n, cell_idx, boundaries, fixed, zero_grad, variables = geo(rng)
        vs = tuple(V.from_str(name) for name, _ in variables)
        B = 3 if tag == "A" else 2
        samples = {name: rng.standard_normal((B, len(cell_idx), d)).astype(np.float32) for name, d in variables}
        # boundary conditions in a fixed iteration order: the FIXED_VALUE ones in `fixed`'s order, the
        # others (no effect on the embedding) interleaved first
        bcs = {}
        for (name, _), v in zip(variables, vs):
            conds = {b: BC(BC.Type.ZERO_GRADIENT) for b in zero_grad.get(name, [])}
            for b, val in fixed.get(name, {}).items():
                conds[b] = BC(BC.Type.FIXED_VALUE, torch.tensor(val))
            bcs[v] = conds
        meta = OpenFOAMMetadata(
            file=Path(f"/tmp/case-{tag}/data.h5"), nu=1e-5, h=np.ones(3), cell_counts=np.array(n),
            cell_idx=torch.tensor(cell_idx), boundaries={k: {"type": "patch", "idx": torch.tensor(i)} for k, i in boundaries.items()},
            boundary_conditions=bcs, holes=[])
        data = OpenFOAMData(meta, torch.zeros(B), {v: torch.tensor(samples[name]) for (name, _), v in zip(variables, vs)})
        x = data.grid_embedding(vs)

        out[f"{tag}/cell_counts"] = np.array(n)
        out[f"{tag}/cell_idx"] = cell_idx
        out[f"{tag}/variables"] = np.array([f"{name}:{d}" for name, d in variables])
        for k, i in boundaries.items():
            out[f"{tag}/boundary/{k}"] = i
        out[f"{tag}/boundary_order"] = np.array(list(boundaries.keys()))
        for name, conds in fixed.items():
            out[f"{tag}/fixed_order/{name}"] = np.array(list(conds.keys()))
            for b, val in conds.items():
                out[f"{tag}/fixed/{name}/{b}"] = val
        for name, s in samples.items():
            out[f"{tag}/samples/{name}"] = s
        out[f"{tag}/grid_embedding"] = to_np(x)

        # statistics as scripts/dataset-stats.py:66-77 stores them (vector fields per component, the
        # norm(...) entries and scalar fields 0-dim)
        stats = {}
        for name, d in variables:
            shape = (d,) if d > 1 else ()
            stats[name] = {"min": -np.abs(rng.standard_normal(shape)).astype(np.float32) - 1,
                           "max": np.abs(rng.standard_normal(shape)).astype(np.float32) + 0.5,
                           "mean": rng.standard_normal(shape).astype(np.float32) * 0.3,
                           "std": np.abs(rng.standard_normal(shape)).astype(np.float32) + 0.2}
            stats[f"norm({name})"] = {"min": np.float32(0.0), "max": np.float32(2.5 + rng.random()),
                                      "mean": np.float32(0.9 + rng.random()), "std": np.float32(0.4 + rng.random())}
        if tag == "B":
            stats["k"]["std"] = np.float32(1e-9)  # exercises the division guard (ofles.py:291)
        for key, st in stats.items():
            for sn, val in st.items():
                out[f"{tag}/stats/{key}/{sn}"] = np.asarray(val)
        ref_stats = OpenFOAMStats({k: {sn: torch.tensor(val) for sn, val in st.items()} for k, st in stats.items()})
        for mi, mode in enumerate(MODES):
            try:  # a per-variable mode must name every variable of the case (KeyError otherwise)
                norm = Normalization(vs, mode)
                mean, std = ref_stats.normalizers(vs, mode)
            except KeyError:
                continue
            out[f"{tag}/mode/{mi}"] = np.array(mode)
            out[f"{tag}/mode/{mi}/mean"] = to_np(mean)
            out[f"{tag}/mode/{mi}/std"] = to_np(std)
            xn = norm.normalize_grid(x, ref_stats)
            out[f"{tag}/mode/{mi}/normalized"] = to_np(xn)
            out[f"{tag}/mode/{mi}/denormalized"] = to_np(norm.denormalize_grid(xn, ref_stats))

        # cell types and their embeddings
        learned = CellTypeEmbedding.create("learned", 4)
        torch.manual_seed(7)
        torch.nn.init.normal_(learned.embedding.weight)
        types = learned.cell_types(data)
        out[f"{tag}/cell_types"] = to_np(types)
        out[f"{tag}/embedding/table"] = to_np(learned.embedding.weight)
        C = learned(data)
        out[f"{tag}/embedding/learned"] = to_np(C)
        gC = torch.tensor(rng.standard_normal(tuple(C.shape)).astype(np.float32))
        C.backward(gC)
        out[f"{tag}/embedding/grad_out"] = to_np(gC)
        out[f"{tag}/embedding/grad_table"] = to_np(learned.embedding.weight.grad)
        out[f"{tag}/embedding/onehot"] = to_np(CellTypeEmbedding.create("onehot", 0)(data))

        # egress: what SampleStore.add_samples writes (metrics.py:52-58) for a denormalised sample
        import einops as eo
        from turbdiff.data.ofles import split_channels

        y = torch.tensor(rng.standard_normal(tuple(x.shape)).astype(np.float32))
        out[f"{tag}/egress/x"] = to_np(y)
        x_v = split_channels(eo.rearrange(select_cells(y, meta.cell_idx), "b f c -> b c f"), vs, dim=-1)
        for (name, _), v in zip(variables, vs):
            out[f"{tag}/egress/{name}"] = to_np(x_v[v].contiguous())

    np.savez_compressed(OUT / "grid_io.npz", **out)
    print(f"wrote {OUT / 'grid_io.npz'}: {(OUT / 'grid_io.npz').stat().st_size / 1e3:.1f} kB, {len(out)} arrays")


if __name__ == "__main__":
    main()
