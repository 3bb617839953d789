"""torch draws stream handles from a pool (32 per device and priority).  After the block backward has made its weight-gradient side
stream (zero-block-only: a 4-KiB scratch arena), how many torch.cuda.Stream() calls until a 'new' stream IS that handle -- and does a
small-grid conv on it still match the main stream's result bit for bit?"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "generative-turbulence_amd"))
import torch
from turbdiff_amd import _lib as L, ops
d = torch.device("cuda:0")
side = torch.cuda.Stream(device=d); L.declare_zero_block_only(side)
B, C, X, Y, Z = 6, 512, 12, 4, 3
g = torch.Generator(device=d).manual_seed(21)
x = torch.randn(B, X, Y, Z, C, device=d, generator=g).bfloat16()
w = torch.randn(C, C, 3, 3, 3, device=d, generator=g) * 0.02
packed = ops._packed_conv3(w, torch.bfloat16)[0]
def conv(y):
    L.call("tdx_conv3_fwd", L.ptr(x), C, None, 0, L.ptr(packed), None, L.ptr(y), B, X, Y, Z, C, L.BF16, L.CONV_AUTO, L.stream())
ref = torch.empty(B, X, Y, Z, C, device=d, dtype=torch.bfloat16); conv(ref); torch.cuda.synchronize()
for n in range(1, 200):
    st = torch.cuda.Stream()
    if st.cuda_stream == side.cuda_stream:
        y = torch.empty_like(ref)
        with torch.cuda.stream(st):
            conv(y)
        torch.cuda.synchronize()
        print(f"stream #{n} created after the side stream has ITS handle; arena {L.scratch_arena(d).numel() if False else 'n/a'}; "
              f"bit-equal to the main stream's result: {torch.equal(y, ref)}; rel-L2 {((y.float() - ref.float()).norm() / ref.float().norm()).item():.2e}")
        break
else:
    print("no collision in 200 streams")
