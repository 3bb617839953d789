import cProfile, pstats, sys, time
from types import SimpleNamespace
sys.path.insert(0, "."); sys.path.insert(0, "generative-turbulence_amd")
import torch, bench
from turbdiff_amd.models.conditioning import Conditioning
from turbdiff_amd.models.ddpm import GaussianDiffusion
from turbdiff_amd.optim import ClipRAdam
dev = torch.device("cuda:0")
diff = GaussianDiffusion(bench.new_cfg1_denoiser(), timesteps=10, beta_schedule="log-snr-linear", loss_type="l2", noise_bcs=True).to(dev)
bench.set_mode(diff, "bf16")
x, c, idx = bench.synthetic_inputs(1, dev, bench.CFG1_GRID)
C = {Conditioning.Type.CELL_TYPE: c}; md = SimpleNamespace(cell_idx=idx)
opt = ClipRAdam(diff.parameters(), lr=1e-4, max_norm=0.1)
def step():
    loss, _ = diff(x, C, md, None); loss.backward(); opt.step(); opt.zero_grad(set_to_none=True)
for _ in range(5): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): step()
torch.cuda.synchronize(); print("ms/step", (time.perf_counter() - t0) * 50)
pr = cProfile.Profile(); pr.enable()
for _ in range(10): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(35)
