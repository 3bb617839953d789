"""The drop-in claim, checked where the reference is available (the build container):
with turbdiff_amd.dropin installed, the reference's own DiffusionTraining (diffusion.py:41-143)
constructs OUR DenoisingModel / GaussianDiffusion and its state_dict has the reference schema.
Skipped on machines without /root/reference (e.g. the GPU box)."""

import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
REF = Path("/root/reference")

SCRIPT = r'''
import sys
sys.path.insert(0, "{root}/tests/golden"); sys.path.insert(0, "{root}/generative-turbulence_amd")
import make_golden
make_golden.install_stubs()
sys.path.insert(0, "/root/reference")
import turbdiff_amd.dropin as dropin
dropin.install()
from pathlib import Path
from turbdiff.models.diffusion import DiffusionTraining          # the reference's Lightning task
from turbdiff.data.ofles import Variable as V
import turbdiff.models.ddpm as D
assert D.__name__ == "turbdiff_amd.models.ddpm", D.__name__
task = DiffusionTraining(Path("/tmp/none"), Path("/tmp/none"), dim=32, variables=(V.U, V.P),
                         beta_schedule="log-snr-linear", timesteps=500, loss="l2", noise_bcs=True,
                         optimizer="radam", norm_type="group", with_geometry_embedding=False)
assert type(task.model).__module__ == "turbdiff_amd.models.ddpm"
assert type(task.model.model).__module__ == "turbdiff_amd.models.ddpm"
want = [l.split("\t")[0] for l in open("{root}/tests/golden/state_dict_manifest.txt") if not l.startswith("#")]
have = list(task.state_dict().keys())
assert have == want, (len(have), len(want), set(have) ^ set(want))
print("DROPIN_OK", len(have))
'''


@pytest.mark.skipif(not REF.exists(), reason="reference checkout not present")
def test_reference_task_builds_our_model():
    out = subprocess.run([sys.executable, "-c", SCRIPT.format(root=ROOT)], capture_output=True, text=True, timeout=300)
    assert "DROPIN_OK 149" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


SURFACE = r'''
import sys, warnings
sys.path.insert(0, "{root}/tests/golden"); sys.path.insert(0, "{root}/generative-turbulence_amd")
import make_golden
make_golden.install_stubs()
sys.path.insert(0, "/root/reference")
import torch
import turbdiff.models.ddpm as R
import turbdiff_amd.models.ddpm as M
import inspect
missing = [n for n, v in vars(R).items() if (inspect.isclass(v) or inspect.isfunction(v))
           and getattr(v, "__module__", "") == R.__name__ and not hasattr(M, n)]
assert not missing, missing
torch.manual_seed(0)
a, b = R.LinearAttention(8, heads=2, dim_head=4), M.LinearAttention(8, heads=2, dim_head=4)
b.load_state_dict(a.state_dict(), strict=True)
x = torch.randn(2, 8, 3, 4, 5)
assert torch.allclose(a(x), b(x), atol=1e-6)
for shape in [(1, 2, 5, 4, 6), (1, 2, 4, 4, 4)]:
    x = torch.randn(*shape)
    (pa, qa), (pb, qb) = R.pad_to_multiple_of(x, 2, mode="constant"), M.pad_to_multiple_of(x, 2, mode="constant")
    assert tuple(qa) == tuple(qb) and torch.equal(pa, pb) and torch.equal(R.unpad(pa, qa), M.unpad(pb, qb))
x, y = torch.randn(2, 3, 4), torch.randn(1, 5, 1)
assert torch.equal(R.expand_as(x, y, 1), M.expand_as(x, y, 1))
# GeometryEmbedding (ddpm.py:375-395; off in the shipped config, kept on stock torch ops)
ga, gb = R.GeometryEmbedding(4, 8, torch.nn.SiLU), M.GeometryEmbedding(4, 8, torch.nn.SiLU)
gb.load_state_dict(ga.state_dict(), strict=True)
cl = torch.randn(4, 55, 46, 47)
assert torch.allclose(ga(cl), gb(cl), atol=1e-6)
# SinusoidalPosEmb, normal_kl, normal_log_lk
assert torch.allclose(R.SinusoidalPosEmb(16)(torch.arange(5.0)), M.SinusoidalPosEmb(16)(torch.arange(5.0)), atol=1e-6)
a, b, c, d = (torch.randn(3, 7) for _ in range(4))
assert torch.allclose(R.normal_kl(a, b, c, d), M.normal_kl(a, b, c, d), atol=1e-6)
assert torch.allclose(R.normal_log_lk(a, b, c), M.normal_log_lk(a, b, c), atol=1e-6)
print("SURFACE_OK")
'''


@pytest.mark.skipif(not REF.exists(), reason="reference checkout not present")
def test_module_surface_matches_reference():
    """Every class / function the reference's ddpm.py defines exists here, and the torch-only ones
    (LinearAttention, pad_to_multiple_of, unpad, expand_as) agree with it numerically."""
    out = subprocess.run([sys.executable, "-c", SURFACE.format(root=ROOT)], capture_output=True, text=True, timeout=300)
    assert "SURFACE_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


GRIDIO = r'''
import sys, inspect
sys.path.insert(0, "{root}/tests/golden"); sys.path.insert(0, "{root}/tests"); sys.path.insert(0, "{root}/generative-turbulence_amd")
import make_golden
make_golden.install_stubs()
sys.path.insert(0, "/root/reference")
import numpy as np, torch
from pathlib import Path
from turbdiff.data.ofles import BoundaryCondition as BC, OpenFOAMData, OpenFOAMMetadata, OpenFOAMStats, Variable as V
import turbdiff.models.normalization as RN, turbdiff.models.cell_type_embeddings as RC, turbdiff.models.conditioning as RCo
import turbdiff_amd.models.normalization as MN, turbdiff_amd.models.cell_type_embeddings as MC, turbdiff_amd.models.conditioning as MCo
import turbdiff_amd.data.ofles as MO, turbdiff.data.ofles as RO
from turbdiff_amd import gridio
from grid_cases import load_case
from test_gridio import emulate_embed

# 1. public names of the reference modules exist here (classes / functions defined in those modules)
for R, M in ((RN, MN), (RC, MC), (RCo, MCo)):
    missing = [n for n, v in vars(R).items() if (inspect.isclass(v) or inspect.isfunction(v))
               and getattr(v, "__module__", "") == R.__name__ and not hasattr(M, n)]
    assert not missing, (R.__name__, missing)
for n in ("Variable", "BoundaryCondition", "split_channels", "OpenFOAMMetadata", "OpenFOAMData", "OpenFOAMStats", "OpenFOAMBatch"):
    assert hasattr(MO, n) and hasattr(RO, n), n
assert [v.name for v in MO.Variable] == [v.name for v in RO.Variable]
assert all(MO.Variable[v.name].dims == v.dims and MO.Variable[v.name].value == v.value for v in RO.Variable)

# 2. the reference's OWN batch objects go through the plan (attribute names only)
for tag in "AB":
    c = load_case(tag)
    vs = tuple(V.from_str(n) for n, _ in c.variables)
    bcs = {{v: {{b: BC(BC.Type.FIXED_VALUE, torch.tensor(val)) for b, val in c.fixed.get(n, {{}}).items()}} for (n, _), v in zip(c.variables, vs)}}
    meta = OpenFOAMMetadata(file=Path("/tmp/x/data.h5"), nu=0.0, h=np.ones(3), cell_counts=np.array(c.cell_counts),
                            cell_idx=torch.tensor(c.cell_idx), boundaries={{k: {{"type": "patch", "idx": torch.tensor(i)}} for k, i in c.boundaries.items()}},
                            boundary_conditions=bcs, holes=[])
    data = OpenFOAMData(meta, torch.zeros(1), {{v: torch.tensor(c.samples[n]) for (n, _), v in zip(c.variables, vs)}})
    ref = data.grid_embedding(vs).numpy()
    plan = gridio.plan_for(data.metadata)
    ours = emulate_embed(plan, plan.features(vs), [c.samples[n] for n, _ in c.variables])
    assert np.array_equal(ours, ref) and np.array_equal(ref, c.grid_embedding)
    emb = RC.CellTypeEmbedding.create("learned", 4)
    assert np.array_equal(emb.cell_types(data).numpy(), plan.types.numpy().reshape(plan.counts))
    # the reference's stats object through OUR Normalization (dense methods) and vice versa
    stats = OpenFOAMStats({{k: {{n: torch.tensor(a) for n, a in st.items()}} for k, st in c.stats.items()}})
    mine = MO.OpenFOAMStats({{k: {{n: torch.tensor(a) for n, a in st.items()}} for k, st in c.stats.items()}})
    mvs = tuple(MO.Variable[v.name] for v in vs)
    for mode in c.modes:
        a = RN.Normalization(vs, mode).normalize_grid(torch.tensor(ref), stats)
        b = MN.Normalization(vs, mode).normalize_grid(torch.tensor(ref), stats)
        m1, s1 = stats.normalizers(vs, mode); m2, s2 = mine.normalizers(mvs, mode)
        assert torch.equal(a, b) and torch.equal(m1, m2) and torch.equal(s1, s2), mode
print("GRIDIO_OK")
'''


@pytest.mark.skipif(not REF.exists(), reason="reference checkout not present")
def test_reference_batch_objects_flow_through_gridio():
    out = subprocess.run([sys.executable, "-c", GRIDIO.format(root=ROOT)], capture_output=True, text=True, timeout=300)
    assert "GRIDIO_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]
