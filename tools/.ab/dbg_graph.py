import sys, os, torch
sys.path.insert(0, "generative-turbulence_amd"); sys.path.insert(0, ".")
from types import SimpleNamespace
import faulthandler; faulthandler.enable()
mode, bump, dim = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
cgrad = int(sys.argv[4]) if len(sys.argv) > 4 else 0
dyn = int(sys.argv[5]) if len(sys.argv) > 5 else 0
from turbdiff_amd.models.ddpm import DenoisingModel, GaussianDiffusion
from turbdiff_amd.models.conditioning import Conditioning
from turbdiff_amd.training import apply_compute_mode
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = DenoisingModel(in_features=4, out_features=4, c_local_features=4, c_global_features=0, timesteps=10, dim=dim, u_net_levels=2, norm_type="group")
apply_compute_mode(net, mode)
diff = GaussianDiffusion(net, timesteps=10, beta_schedule="log-snr-linear", loss_type="l2", noise_bcs=True).to(dev)
X, Y, Z = 12, 10, 9
x = torch.randn(2, 4, X, Y, Z, device=dev); c = torch.randn(4, X, Y, Z, device=dev)
m = torch.zeros(X, Y, Z, dtype=torch.bool); m[1:-1, 1:-1, 1:-1] = True
idx = m.flatten().nonzero().flatten().to(dev)
if cgrad: c.requires_grad_()
C = {Conditioning.Type.CELL_TYPE: c}; md = SimpleNamespace(cell_idx=idx)
if dyn:
    mask, n = diff.domain_mask(idx, X * Y * Z)
    md = SimpleNamespace(cell_idx=None, domain_mask=(mask.clone(), torch.tensor([n], dtype=torch.int64, device=dev)))
params = list(diff.parameters()) + ([c] if cgrad else [])
def step():
    loss, _ = diff(x, C, md, None); loss.backward(); return loss
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        for p in params: p.grad = None
        step()
torch.cuda.current_stream().wait_stream(s)
for p in params: p.grad = None
if bump: torch.autograd.graph.increment_version(params)
g = torch.cuda.CUDAGraph()
print("capturing", mode, bump, dim, flush=True)
with torch.cuda.graph(g, stream=s):
    loss = step()
print("captured", flush=True)
g.replay(); torch.cuda.synchronize(); print("replayed", loss.item(), flush=True)
