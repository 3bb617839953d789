"""Data ingress / egress of the hot path on the GPU: host side of ``tdx_grid_embed``,
``tdx_grid_select`` and ``tdx_cell_embed_*`` (csrc/tdx_gridio.hip, ABI in include/tdx.h).

What the reference does per batch with a chain of index_put / addcmul / embedding calls
(``OpenFOAMData.grid_embedding`` ofles.py:220-240, ``Normalization`` normalization.py:19-29,
``CellTypeLearnedEmbedding`` cell_type_embeddings.py:62-70, ``select_cells`` utils.py:14-15 as used by
``SampleStore.add_samples`` metrics.py:52-58) becomes one kernel each.  The per-geometry part -- which
voxel holds which cell, which boundary value wins where, the cell-type grid -- is resolved ONCE per
geometry into a ``GridPlan`` (plain torch index ops in the reference's write order, device-agnostic, so it
is testable without a GPU); the kernels then only stream.

The functions read attribute names only (``data.samples``, ``data.metadata.{cell_counts, cell_idx,
boundaries[name]["idx"], boundary_conditions[v][name].{type, value}}``), so the reference's own
``OpenFOAMData`` / ``OpenFOAMBatch`` objects work as well as ``turbdiff_amd.data.ofles``'s.
There is no CPU path: dense results need the HIP library and device tensors.
"""

from __future__ import annotations

import torch

from . import _lib as L

CELL_TYPES = {"inside": 0, "outside": 1, "walls": 2, "inlets": 3, "outlets": 4, "empties": 5}  # cell_type_embeddings.py:30-38
MAX_VARS = 4


def _name(v) -> str:
    return v.name.lower()


def _is_fixed(desc) -> bool:
    return getattr(desc.type, "name", str(desc.type)) == "FIXED_VALUE"


class FeaturePlan:
    """FIXED_VALUE boundary values of one variable tuple on one geometry, in the reference's write order
    (ofles.py:235-238: per variable, its boundary conditions in dict order; later writes win).

    ovr_of (V,) int32 row of the override table or -1 (None if there are no rows); ovr_val (R, F) f32;
    ovr_mask (R,) int32, bit f set = feature f is fixed at that voxel."""

    def __init__(self, metadata, variables, V, device):
        self.dims = [v.dims for v in variables]
        self.F = F = sum(self.dims)
        if len(variables) > MAX_VARS or F > 32:
            raise RuntimeError(f"at most {MAX_VARS} variables / 32 features, got {len(variables)} / {F}")
        val = torch.zeros((F, V), dtype=torch.float32, device=device)
        fixed = torch.zeros((F, V), dtype=torch.bool, device=device)
        f0 = 0
        for v, d in zip(variables, self.dims):
            for bname, desc in metadata.boundary_conditions.get(v, {}).items():
                if _is_fixed(desc):
                    idx = metadata.boundaries[bname]["idx"].long().to(device)
                    value = torch.as_tensor(desc.value, dtype=torch.float32, device=device).reshape(-1)
                    val[f0:f0 + d, idx] = value.expand(d)[:, None] if value.numel() == 1 else value[:, None]
                    fixed[f0:f0 + d, idx] = True
            f0 += d
        rows = fixed.any(dim=0).nonzero().flatten()
        self.ovr_of = self.ovr_val = self.ovr_mask = None
        if rows.numel():
            ovr_of = torch.full((V,), -1, dtype=torch.int32, device=device)
            ovr_of[rows] = torch.arange(rows.numel(), dtype=torch.int32, device=device)
            bits = (1 << torch.arange(F, dtype=torch.int64, device=device))[:, None]
            mask = (fixed[:, rows].long() * bits).sum(dim=0)
            # int32 storage of a 32-bit mask (bit 31 -> sign)
            self.ovr_mask = torch.where(mask >= 2**31, mask - 2**32, mask).to(torch.int32).contiguous()
            self.ovr_of, self.ovr_val = ovr_of, val[:, rows].t().contiguous()


class GridPlan:
    """Per-geometry constants of the ingress / egress kernels.

    cell_of (V,) int32   position of voxel v in the cell list, -1 outside the domain
    types   (V,) uint8   cell types (cell_type_embeddings.py:47-59)
    features(variables)  the FIXED_VALUE override table of a variable tuple (cached)
    """

    def __init__(self, metadata):
        device = metadata.cell_idx.device
        self.counts = tuple(int(c) for c in metadata.cell_counts)
        V = 1
        for c in self.counts:
            V *= c
        if V >= 2**31:
            raise RuntimeError("grids of 2^31 voxels or more are not supported")
        self.V, self.metadata = V, metadata
        cell_idx = metadata.cell_idx.long().contiguous()
        self.cell_idx, self.n_cells = cell_idx, cell_idx.numel()
        cell_of = torch.full((V,), -1, dtype=torch.int32, device=device)
        cell_of[cell_idx] = torch.arange(self.n_cells, dtype=torch.int32, device=device)
        self.cell_of = cell_of
        types = torch.full((V,), CELL_TYPES["outside"], dtype=torch.uint8, device=device)
        types[cell_idx] = CELL_TYPES["inside"]
        for bname, desc in metadata.boundaries.items():
            types[desc["idx"].long().to(device)] = CELL_TYPES[bname]
        self.types = types
        self._features = {}

    def features(self, variables) -> FeaturePlan:
        key = tuple(_name(v) for v in variables)
        if key not in self._features:
            self._features[key] = FeaturePlan(self.metadata, variables, self.V, self.cell_of.device)
        return self._features[key]


def plan_for(metadata) -> GridPlan:
    """Cached on the metadata object (the reference caches the embedding itself per batch object)."""
    plan = metadata.__dict__.get("_tdx_plan")
    if plan is None or plan.cell_of.device != metadata.cell_idx.device:
        plan = metadata.__dict__["_tdx_plan"] = GridPlan(metadata)
    return plan


def _slots(tensors, dims):
    args = []
    for i in range(MAX_VARS):
        if i < len(tensors):
            args += [L.ptr(tensors[i]), dims[i]]
        else:
            args += [None, 0]
    return args


_AFFINE: dict = {}


def _shift_scale(mean, std, device):
    """(-mean/std, 1/std) on the device, cached per (mean, std) tensor pair (OpenFOAMStats caches those)."""
    key = (id(mean), id(std), str(device))
    hit = _AFFINE.get(key)
    if hit is None or hit[0] is not mean or hit[1] is not std:
        m, s = mean.to(device, torch.float32), std.to(device, torch.float32)
        if len(_AFFINE) > 64:
            _AFFINE.clear()
        hit = _AFFINE[key] = (mean, std, (-m / s).contiguous(), torch.reciprocal(s).contiguous(), m.contiguous(), s.contiguous())
    return hit[2], hit[3]


def _mean_std(mean, std, device):
    _shift_scale(mean, std, device)
    hit = _AFFINE[(id(mean), id(std), str(device))]
    return hit[4], hit[5]


def grid_embed(data, variables, mean=None, std=None):
    """Dense (B, F, X, Y, Z) fp32 grid of a batch; with ``mean`` / ``std`` (F,) also normalised as
    ``Normalization.normalize_grid`` does: addcmul(-mean/std, 1/std, x)."""
    plan = plan_for(data.metadata)
    fp = plan.features(variables)
    samples = []
    for v, d in zip(variables, fp.dims):
        s = data.samples[v]
        if s.dtype != torch.float32 or s.shape[-2:] != (plan.n_cells, d) or s.ndim != 3:
            raise RuntimeError(f"samples of {v} must be fp32 (B, {plan.n_cells}, {d}), got {s.dtype} {tuple(s.shape)}")
        samples.append(s.contiguous())
    B = samples[0].shape[0]
    x = torch.empty((B, fp.F, *plan.counts), dtype=torch.float32, device=samples[0].device)
    shift = scale = None
    if mean is not None:
        shift, scale = _shift_scale(mean, std, x.device)
    L.call("tdx_grid_embed", *_slots(samples, fp.dims), L.ptr(plan.cell_of), L.ptr(fp.ovr_of), L.ptr(fp.ovr_val),
           L.ptr(fp.ovr_mask), L.ptr(shift), L.ptr(scale), L.ptr(x), B, plan.n_cells, plan.V, L.stream())
    return x


def grid_select(x, metadata, variables, mean=None, std=None):
    """{variable: (B, n_cells, dims)} channels-last in-domain values of a dense (B, F, X, Y, Z) fp32 grid;
    with ``mean`` / ``std`` denormalised first (addcmul(mean, std, x))."""
    plan = plan_for(metadata)
    dims = [v.dims for v in variables]
    if x.dtype != torch.float32 or x.ndim != 5 or tuple(x.shape[1:]) != (sum(dims), *plan.counts):
        raise RuntimeError(f"expected fp32 (B, {sum(dims)}, {plan.counts}), got {x.dtype} {tuple(x.shape)}")
    x = x.contiguous()
    B = x.shape[0]
    outs = [torch.empty((B, plan.n_cells, d), dtype=torch.float32, device=x.device) for d in dims]
    if mean is not None:
        mean, std = _mean_std(mean, std, x.device)
    L.call("tdx_grid_select", L.ptr(x), L.ptr(plan.cell_idx), L.ptr(mean), L.ptr(std), *_slots(outs, dims), B,
           plan.n_cells, plan.V, L.stream())
    return dict(zip(variables, outs))


class _CellEmbed(torch.autograd.Function):
    @staticmethod
    def forward(ctx, table, types, counts):
        n_types, D = table.shape
        out = torch.empty((D, *counts), dtype=torch.float32, device=table.device)
        L.call("tdx_cell_embed_fwd", L.ptr(types), L.ptr(table.detach().float().contiguous()), L.ptr(out), n_types, D,
               types.numel(), L.stream())
        ctx.save_for_backward(types)
        ctx.shape = (n_types, D)
        return out

    @staticmethod
    def backward(ctx, dC):
        (types,) = ctx.saved_tensors
        n_types, D = ctx.shape
        dC = dC.contiguous().float()
        dtable = torch.empty((n_types, D), dtype=torch.float32, device=dC.device)
        ws = torch.empty(L.query("tdx_cell_embed_bwd_workspace_bytes", n_types, D), dtype=torch.uint8, device=dC.device)
        L.call("tdx_cell_embed_bwd", L.ptr(types), L.ptr(dC), L.ptr(dtable), 0, n_types, D, types.numel(), L.ptr(ws),
               L.stream())
        return dtable, None, None


def cell_type_embedding(table, plan: GridPlan):
    """(D, X, Y, Z) = movedim(table[types], -1, 0), differentiable w.r.t. ``table`` (n_types, D)."""
    return _CellEmbed.apply(table, plan.types, plan.counts)
