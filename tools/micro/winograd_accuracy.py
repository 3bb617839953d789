#!/usr/bin/env python3
"""Pricing, part 1 (CPU, no GPU): what a Winograd-domain contraction along z -- F(2, 3): two outputs from four transformed
taps, 1.5x fewer multiplies; then F(2x2, 3x3) along y and z: 2.25x -- does to the ERROR of a 3x3x3 conv (reference
ddpm.py:164) when the matrix-core operands are 16-bit.  Emulated exactly as a kernel would compute it: inputs and weights
rounded to the operand format, the input transform B^T d done IN that format (packed 16-bit adds between ds_read and MFMA:
the only place a kernel can afford it), the weight transform G g done once in fp32 and rounded, products and sums in
fp32 (MFMA accumulators), output transform A^T m in fp32.  Reference: the fp64 conv of the SAME rounded inputs and
weights (so that only the arithmetic differs), zero padding (the transform does not care), 64 -> 64 channels.
    python tools/micro/winograd_accuracy.py"""
import torch
import torch.nn.functional as F

torch.manual_seed(0)
Cin, Cout, X, Y, Z = 64, 64, 10, 12, 16


def conv_direct(x, w):
    return F.conv3d(x, w, padding=1)


def winograd_z(x, w, dt, also_y=False):
    """x (1, Cin, X, Y, Z) and w (Cout, Cin, 3, 3, 3) already representable in dt."""
    G = torch.tensor([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]], dtype=torch.float32)
    xp = F.pad(x, (1, 1, 1, 1, 1, 1))  # zero padding on all axes
    rnd = lambda t: t.to(dt).float()
    # weight transform in fp32, then rounded to the operand format: U[i] = sum_k G[i, k] w[..., k]
    U = rnd(torch.einsum("ik,oc x y k->i o c x y".replace(" ", ""), G, w.float()))
    if also_y:
        U = rnd(torch.einsum("jk,iocxk->ijocx", G, torch.einsum("ik,ocxyk->iocxy", G, w.float()).permute(0, 1, 2, 3, 4)))
    zp = Z // 2
    # input transform along z in the operand format: d = xp[..., 2r : 2r + 4]
    d = [xp[..., j : j + 2 * zp : 2] for j in range(4)]  # each (1, Cin, X+2, Y+2, zp)
    sub = lambda a, b: rnd(a - b)  # one packed 16-bit add, rounded
    V = [sub(d[0], d[2]), rnd(d[1] + d[2]), sub(d[2], d[1]), sub(d[1], d[3])]
    if not also_y:
        # m_i = sum over (dx, dy, c) of U_i[dx, dy] V_i shifted: a 2-D (x, y) conv per z-pair column, fp32 accumulation
        m = []
        for i in range(4):
            vi = V[i].permute(0, 4, 1, 2, 3).reshape(zp, Cin, X + 2, Y + 2)  # z-pairs as batch
            m.append(F.conv2d(vi, U[i]))  # (zp, Cout, X, Y)
        y0, y1 = m[0] + m[1] + m[2], m[1] - m[2] - m[3]
        out = torch.stack((y0, y1), dim=-1)  # (zp, Cout, X, Y, 2)
        return out.permute(1, 2, 3, 0, 4).reshape(1, Cout, X, Y, Z)
    # F(2x2, 3x3) over (y, z): transform along y as well (a second packed-add stage in the operand format)
    yp = Y // 2
    Vy = []
    for i in range(4):
        e = [V[i][..., j : j + 2 * yp : 2, :] for j in range(4)]
        Vy.append([sub(e[0], e[2]), rnd(e[1] + e[2]), sub(e[2], e[1]), sub(e[1], e[3])])
    m = [[None] * 4 for _ in range(4)]
    for i in range(4):       # z index
        for j in range(4):   # y index
            v = Vy[i][j].permute(0, 3, 4, 1, 2).reshape(yp * zp, Cin, X + 2)
            m[i][j] = F.conv1d(v, U[i, j])  # U[i, j]: (Cout, Cin, 3) over x -> (yp*zp, Cout, X)
    AT = [[1, 1, 1, 0], [0, 1, -1, -1]]
    out = torch.zeros(yp, zp, Cout, X, 2, 2)
    for a in range(2):
        for b in range(2):
            acc = 0
            for i in range(4):
                for j in range(4):
                    if AT[b][i] and AT[a][j]:
                        acc = acc + AT[b][i] * AT[a][j] * m[i][j]
            out[..., a, b] = acc.reshape(yp, zp, Cout, X)
    return out.permute(2, 3, 0, 4, 1, 5).reshape(1, Cout, X, Y, Z)


rel = lambda a, b: ((a.double() - b).norm() / b.norm()).item()
print(f"3x3x3 conv {Cin} -> {Cout} on {X}x{Y}x{Z}, N(0,1) activations, N(0, 2/(27 Cin)) weights; rel-L2 against the fp64 conv of the same rounded operands")
for name, dt in (("bf16", torch.bfloat16), ("fp16", torch.float16)):
    x = torch.randn(1, Cin, X, Y, Z).to(dt).float()
    w = (torch.randn(Cout, Cin, 3, 3, 3) * (2.0 / (27 * Cin)) ** 0.5).to(dt).float()
    ref = conv_direct(x.double(), w.double())
    direct = conv_direct(x, w)  # exact products of 16-bit operands, fp32 sums: what the MFMA kernels compute
    # what a stored result costs: rounding the fp32 result to the storage format
    store = rel(direct.to(dt), ref)
    wz = winograd_z(x, w, dt)
    wyz = winograd_z(x, w, dt, also_y=True)
    print(f"{name}: direct {rel(direct, ref):.2e} (stored in {name}: {store:.2e}) | Winograd z F(2,3) {rel(wz, ref):.2e} "
          f"(stored {rel(wz.to(dt), ref):.2e}) | Winograd y-z F(2x2,3x3) {rel(wyz, ref):.2e} (stored {rel(wyz.to(dt), ref):.2e})")
