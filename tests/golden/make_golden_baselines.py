#!/usr/bin/env python3
"""Golden vectors of the baseline conv layers, from the reference's own classes.

Build container only: imports the unmodified ``turbdiff.models.dilresnet.DilatedCNNBlock`` and
``turbdiff.models.tfnet.conv`` / ``deconv`` (stand-ins for uninstalled third-party packages as in make_golden.py),
runs them forward and backward on seeded inputs and stores inputs, parameters, outputs and all gradients in
tests/golden/baselines.npz.

    python tests/golden/make_golden_baselines.py
"""

import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent))
from make_golden import REF, install_stubs, to_np  # noqa: E402

OUT = Path(__file__).resolve().parent


def run(tag, module, x, out):
    sd0 = {k: v.clone() for k, v in module.state_dict().items()}  # before the forward updates running statistics
    x = x.clone().requires_grad_()
    y = module(x)
    gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(99))
    y.backward(gy)
    out[f"{tag}/x"], out[f"{tag}/y"], out[f"{tag}/gy"], out[f"{tag}/gx"] = to_np(x), to_np(y), to_np(gy), to_np(x.grad)
    for k, v in sd0.items():
        out[f"{tag}/sd/{k}"] = to_np(v)
    for k, p in module.named_parameters():
        out[f"{tag}/grad/{k}"] = to_np(p.grad)


def main():
    install_stubs()
    sys.path.insert(0, str(REF))
    torch.set_num_threads(8)
    from turbdiff.models.dilresnet import DilatedCNNBlock
    from turbdiff.models import tfnet

    g = torch.Generator().manual_seed(5)
    out = {}
    torch.manual_seed(11)
    blk = DilatedCNNBlock(16, [1, 2, 4])           # dilations 1, 2, 4, 2, 1 on a grid smaller than the largest halo
    run("dil", blk, torch.randn(2, 16, 9, 7, 6, generator=g), out)
    out["dil/dilations"] = np.array([1, 2, 4])
    torch.manual_seed(12)
    blk8 = DilatedCNNBlock(8, [8])                 # one dilation-8 layer
    run("dil8", blk8, torch.randn(1, 8, 11, 5, 20, generator=g), out)
    out["dil8/dilations"] = np.array([8])
    torch.manual_seed(13)
    cv = tfnet.conv(8, 16, kernel_size=3, stride=2, dropout_rate=0.0)
    with torch.no_grad():  # the reference's Encoder re-initialises to N(0, 0.002 / n): use weights of a visible size
        cv[0].weight.normal_(0, 0.2); cv[0].bias.normal_(0, 0.1); cv[1].weight.uniform_(0.5, 1.5); cv[1].bias.normal_(0, 0.1)
    cv.train()
    run("conv_s2", cv, torch.randn(2, 8, 9, 8, 7, generator=g), out)   # odd and even extents
    torch.manual_seed(14)
    cv5 = tfnet.conv(8, 8, kernel_size=5, stride=2, dropout_rate=0.0)
    with torch.no_grad():
        cv5[0].weight.normal_(0, 0.1); cv5[0].bias.normal_(0, 0.1)
    cv5.eval()                                                        # running statistics (0, 1)
    run("conv_k5", cv5, torch.randn(1, 8, 10, 6, 7, generator=g), out)
    torch.manual_seed(15)
    dc = tfnet.deconv(16, 8)
    with torch.no_grad():
        dc[0].weight.normal_(0, 0.2); dc[0].bias.normal_(0, 0.1)
    run("deconv", dc, torch.randn(2, 16, 5, 4, 3, generator=g), out)
    np.savez_compressed(OUT / "baselines.npz", **out)
    print("wrote", OUT / "baselines.npz", f"{(OUT / 'baselines.npz').stat().st_size / 1e3:.0f} kB")


if __name__ == "__main__":
    main()
