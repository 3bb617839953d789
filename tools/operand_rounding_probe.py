"""What would modes BETWEEN fp16 (fp16 tensors + fp16 MFMA) and f32s (fp32 tensors, three bf16 products) buy in accuracy?
Emulated on the CPU oracle (bench.py's accuracy probe: eps-hat at t = 250, B = 1, 192x64x48, default init, rel-L2 against the
plain fp32 oracle): the 3x3x3 / 1x1 convs' operands rounded to a 16-bit format before an fp32 convolution, tensors left in fp32.
  tf32-like : activations AND weights rounded to fp16 (11 significand bits)  -- the reference's TF32 arithmetic, one MFMA pass
  two-pass  : weights rounded to fp16, activations exact                      -- two passes (a_hi w_hi + a_lo w_hi)
  bf16-ops  : both rounded to bf16, fp32 tensors                              -- what bf16 MFMA costs in accuracy without bf16 storage
Runs in ~1 min on 8 cores:  python tools/operand_rounding_probe.py"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "generative-turbulence_amd"))
import torch
import torch.nn.functional as F

import bench
from oracle import turbdiff_oracle as O

torch.set_num_threads(8)
net = bench.new_denoiser()
sd = {k: v.clone() for k, v in net.state_dict().items()}
x, c, cell_idx = bench.synthetic_inputs(1, "cpu")
buf = O.schedule_buffers("log-snr-linear", 500)
t = torch.tensor([bench.ACCURACY_T])
noise = torch.randn(x.shape, generator=torch.Generator().manual_seed(1))
x_t = O.q_sample(buf, x, t, noise)
real_conv3d = F.conv3d


def rounded(act_dtype, w_dtype):
    def conv3d(inp, w, *a, **k):
        if act_dtype is not None:
            inp = inp.to(act_dtype).float()
        if w_dtype is not None:
            w = w.to(w_dtype).float()
        return real_conv3d(inp, w, *a, **k)

    return conv3d


def eps_hat():
    with torch.no_grad():
        return O.denoiser(sd, x_t, t, c, timesteps=500)


ref = eps_hat()
for name, (a, w) in {"tf32-like (both fp16)": (torch.float16, torch.float16), "two-pass (weights fp16)": (None, torch.float16),
                     "activations fp16 only": (torch.float16, None), "bf16 operands, fp32 tensors": (torch.bfloat16, torch.bfloat16)}.items():
    F.conv3d = rounded(a, w)
    try:
        e = eps_hat()
    finally:
        F.conv3d = real_conv3d
    print(f"{name:30s} rel-L2 vs the fp32 oracle {((e - ref).norm() / ref.norm()).item():.3e}", flush=True)
