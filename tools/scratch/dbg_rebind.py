import sys; sys.path.insert(0,'.'); sys.path.insert(0,'generative-turbulence_amd')
import torch, bench
from types import SimpleNamespace
from turbdiff_amd.models.conditioning import Conditioning
from turbdiff_amd.models.ddpm import GaussianDiffusion
import turbdiff_amd.models.ddpm as D
dev=torch.device('cuda:0')
diff = GaussianDiffusion(bench.new_cfg1_denoiser(), timesteps=10, beta_schedule="log-snr-linear", loss_type="l2", noise_bcs=True).to(dev)
x, c, cell_idx = bench.synthetic_inputs(1, dev, bench.CFG1_GRID)
C = {Conditioning.Type.CELL_TYPE: c}
bench.set_mode(diff,'bf16')
orig = diff.model._encode_local
def dbg(Cc):
    e = orig(Cc)
    print('encode_local: grad', torch.is_grad_enabled(), 'dtype', e.dtype, 'partial', getattr(e,'first_conv_partial',None) is not None, 'impl', D._lib.conv_impl())
    return e
diff.model._encode_local = dbg
diff.p_sample_loop(x, C, cell_idx)
diff.p_sample_loop(x, C, cell_idx)
print('ok')
