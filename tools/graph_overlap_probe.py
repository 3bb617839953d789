"""Run under `rocprofv3 --kernel-trace`: 8 eager training steps, a marker kernel, 8 captured steps (bench.py's headline
workload).  tools/graph_overlap_report.py reads the trace and says how much of the weight-gradient side stream's work
overlaps the data-gradient chain in each half."""
import sys
from pathlib import Path
from types import SimpleNamespace

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "generative-turbulence_amd"))
import torch

import bench
from turbdiff_amd.models.conditioning import Conditioning
from turbdiff_amd.training import GraphedTrainingStep

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 6
diff = bench.build_model(dev)
x, c_local, cell_idx = bench.synthetic_inputs(B, dev)
C = {Conditioning.Type.CELL_TYPE: c_local}
md = SimpleNamespace(cell_idx=cell_idx)
bench.set_mode(diff, "bf16")
opt = bench.new_optimizer(diff, "bf16", bench.LOSS_ELEMENTS(B, cell_idx))
marker = torch.arange(1024, device=dev, dtype=torch.float32)


def eager():
    loss, _ = diff(x, C, md, None)
    loss.backward()
    opt.step()
    opt.zero_grad(set_to_none=True)


for _ in range(8):
    eager()
torch.cuda.synchronize()
torch.cumsum(marker, 0)
torch.cuda.synchronize()
task = bench._Task(diff, opt)
gs = GraphedTrainingStep(task)
batch = SimpleNamespace(x=x, C=C, cell_idx=cell_idx)
diff.zero_grad(set_to_none=True)
for _ in range(8):
    gs(batch)
    opt.step()
torch.cuda.synchronize()
