"""debug: run-to-run noise of parameter gradients vs the deferred-encode difference (bf16)"""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "generative-turbulence_amd"))
from turbdiff_amd.models import ddpm as D
from turbdiff_amd.models.conditioning import Conditioning
dev = torch.device("cuda")
torch.manual_seed(0)
net = D.DenoisingModel(in_features=4, out_features=4, c_local_features=4, c_global_features=0, timesteps=500, dim=32,
                       u_net_levels=2, norm_type="group")
x = torch.randn(3, 4, 26, 12, 17, generator=torch.Generator().manual_seed(1)).to(dev)
c_local = torch.randn(4, 26, 12, 17, generator=torch.Generator().manual_seed(2)).to(dev)
t = torch.tensor([3, 250, 499]).to(dev)
net.to(dev).set_compute_dtype(torch.bfloat16)
C = {Conditioning.Type.CELL_TYPE: c_local}
gy = torch.randn(3, 4, 26, 12, 17, generator=torch.Generator().manual_seed(3)).to(dev)
def run(defer):
    D.DEFER_ENCODE = defer
    net.zero_grad(set_to_none=True)
    y = net(x, t, C)
    (y * gy).sum().backward()
    return {n: p.grad.clone() for n, p in net.named_parameters()}
rel = lambda a, b: float((a - b).norm() / b.norm().clamp_min(1e-30))
a, a2, b, b2 = run(True), run(True), run(False), run(False)
for n in a:
    r = (rel(a[n], a2[n]), rel(b[n], b2[n]), rel(a[n], b[n]))
    if max(r) > 1e-6:
        print(f"{n:50s} defer/defer {r[0]:.2e}  plain/plain {r[1]:.2e}  defer/plain {r[2]:.2e}")
