"""debug: how tdx_gn_apply_encoded differs from encode + gn_apply in bf16"""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "generative-turbulence_amd"))
from turbdiff_amd import _lib as L, ops
d = torch.device("cuda")
g = torch.Generator().manual_seed(0)
rnd = lambda *s: torch.randn(*s, generator=g)
B, D, G, grid = 2, 32, 8, (13, 9, 11)
V = grid[0] * grid[1] * grid[2]
C = 2 * D
dtype = torch.bfloat16
x, c = rnd(B, 4, *grid).to(d), rnd(4, *grid).to(d)
wx, bx, wc, bc = rnd(D, 4, 1, 1, 1).to(d), rnd(D).to(d), rnd(D, 4, 1, 1, 1).to(d), rnd(D).to(d)
h2 = rnd(B, *grid, C).to(d).to(dtype)
stats = torch.stack((rnd(B, G) * 0.1, rnd(B, G).abs() + 0.5), dim=-1).contiguous().to(d)
gamma, beta = rnd(C).to(d), rnd(C).to(d)
code = L.dtype_code(dtype)
res = ops.encode(x, c, wx, bx, wc, bc, dtype)
ref = torch.empty_like(h2)
L.call("tdx_gn_apply", L.ptr(h2), L.ptr(stats), L.ptr(gamma), L.ptr(beta), None, None, L.ptr(res), L.ptr(ref), B, V, C, G, 1, code, L.stream())
out = torch.empty_like(h2)
wx2, wc2 = wx.reshape(D, 4).contiguous(), wc.reshape(D, 4).contiguous()
L.call("tdx_gn_apply_encoded", L.ptr(h2), L.ptr(stats), L.ptr(gamma), L.ptr(beta), L.ptr(x), 4, L.ptr(wx2), L.ptr(bx), L.ptr(c), 4,
       L.ptr(wc2), L.ptr(bc), L.ptr(out), B, V, D, G, code, L.stream())
diff = (out.float() - ref.float())
nz = diff != 0
print("elements", diff.numel(), "differing", int(nz.sum()), "max abs", float(diff.abs().max()))
idx = nz.nonzero()[:10]
for i in idx:
    i = tuple(int(k) for k in i)
    print(i, float(out[i]), float(ref[i]), "res", float(res[i]), "h2", float(h2[i]))
# zero the GN part: gamma = 0, beta = 0 -> silu(0) = 0 -> output = the skip itself
z = torch.zeros_like(gamma)
L.call("tdx_gn_apply_encoded", L.ptr(h2), L.ptr(stats), L.ptr(z), L.ptr(z), L.ptr(x), 4, L.ptr(wx2), L.ptr(bx), L.ptr(c), 4,
       L.ptr(wc2), L.ptr(bc), L.ptr(out), B, V, D, G, code, L.stream())
d2 = out.float() - res.float()
print("skip alone: differing", int((d2 != 0).sum()), "max abs", float(d2.abs().max()))
