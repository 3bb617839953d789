"""Accuracy report at BASELINE configs[1] size (a checker script, kept under tests/ because it runs the CPU oracle):
rel-L2 of the HIP forward against the oracle at 192x64x48 in the fp32 / split-precision / bf16 modes.
GPU box:  python tests/report_accuracy.py
"""
import sys, torch
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "generative-turbulence_amd")); sys.path.insert(0, str(ROOT / "tests"))
from oracle import turbdiff_oracle as O
from turbdiff_amd.models.ddpm import DenoisingModel
from turbdiff_amd.models.conditioning import Conditioning
from conftest import rel_l2
torch.manual_seed(0)
net = DenoisingModel(in_features=4, out_features=4, c_local_features=4, c_global_features=0, timesteps=500, dim=32, u_net_levels=4, norm_type="group")
sd = {k: v.clone() for k, v in net.state_dict().items()}
x = torch.randn(1, 4, 192, 64, 48, generator=torch.Generator().manual_seed(1234))
c = torch.randn(4, 192, 64, 48, generator=torch.Generator().manual_seed(1235))
t = torch.tensor([123])
d = torch.device("cuda:0")
C = {Conditioning.Type.CELL_TYPE: c.to(d)}
with torch.no_grad():
    ref = O.denoiser(sd, x, t, c, timesteps=500)
    net.to(d)
    y32 = net(x.to(d), t.to(d), C)
    import os
    os.environ["TDX_CONV_IMPL"] = "split"   # fp32 tensors, split-precision (bf16 hi + lo) MFMA convs
    y32s = net(x.to(d), t.to(d), C)
    del os.environ["TDX_CONV_IMPL"]
    net.set_compute_dtype(torch.bfloat16)
    y16 = net(x.to(d), t.to(d), C)
    y16c = net(x.to(d), t.to(d), C, encoded_local=net.encode_local(C))
print("rel-L2 vs CPU oracle at 192x64x48: fp32 mode %.2e, fp32 tensors + split-precision convs %.2e, bf16 mode %.2e, bf16 + cached conditioning conv %.2e" % (rel_l2(y32.cpu(), ref), rel_l2(y32s.cpu(), ref), rel_l2(y16.cpu(), ref), rel_l2(y16c.cpu(), ref)))
