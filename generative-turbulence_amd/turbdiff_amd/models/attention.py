"""``fused_attention(q, k, v)`` with the reference's signature (turbdiff/models/attention.py:9-15).

The reference calls ``F.scaled_dot_product_attention`` on three (b, h, n, d) tensors that it
first permutes out of the to_qkv conv output.  The HIP kernel works on the token-major q|k|v
layout directly, so this wrapper only re-packs its arguments; the U-Net itself calls
``ops.attention`` on the conv output without any copy.
"""

import torch

from .. import ops


def fused_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    b, h, n, d = q.shape
    pack = lambda t: t.transpose(1, 2).reshape(b, n, h * d)
    qkv = torch.cat((pack(q), pack(k), pack(v)), dim=-1)
    out = ops.attention(qkv, h)  # (b, n, h*d)
    return out.reshape(b, n, h, d).transpose(1, 2)
