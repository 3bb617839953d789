"""CPU oracle of the baseline conv layers -- TEST INFRASTRUCTURE, never imported by the product.

Functional restatements on stock PyTorch-CPU ops of
    DilatedCNNBlock.forward          turbdiff/models/dilresnet.py:40-44   (layers built at :28-37)
    tfnet.conv(...) / tfnet.deconv(...) layer stacks    turbdiff/models/tfnet.py:185-208
on plain state dicts.  PINNED by tests/golden/baselines.npz, which tests/golden/make_golden_baselines.py generates by
running the reference's own classes (tests/test_oracle_golden.py::test_baseline_conv_oracle)."""

import torch
import torch.nn.functional as F


def dilated_block(sd, x, dilations, pre="layers."):
    """x (B, C, X, Y, Z); layers.i = Conv3d(3, dilation d_i, replicate padding d_i) then ReLU, d = d1..dn..d1."""
    ds = list(dilations) + list(reversed(dilations[:-1]))
    for i, d in enumerate(ds):
        xp = F.pad(x, (d,) * 6, mode="replicate")
        x = F.relu(F.conv3d(xp, sd[f"{pre}{i}.weight"], sd[f"{pre}{i}.bias"], dilation=d))
    return x


def tfnet_conv(sd, x, kernel_size, stride, training=True, pre=""):
    """Conv3d(k, stride, zero padding (k-1)//2) -> BatchNorm3d (batch statistics when training) -> LeakyReLU(0.1);
    dropout rate 0 in the fixtures."""
    h = F.conv3d(x, sd[pre + "0.weight"], sd[pre + "0.bias"], stride=stride, padding=(kernel_size - 1) // 2)
    h = F.batch_norm(h, sd[pre + "1.running_mean"].clone(), sd[pre + "1.running_var"].clone(), sd[pre + "1.weight"], sd[pre + "1.bias"],
                     training, 0.1, 1e-5)
    return F.leaky_relu(h, 0.1)


def tfnet_deconv(sd, x, pre=""):
    """ConvTranspose3d(4, stride 2, padding 1) -> LeakyReLU(0.1)."""
    return F.leaky_relu(F.conv_transpose3d(x, sd[pre + "0.weight"], sd[pre + "0.bias"], stride=2, padding=1), 0.1)
