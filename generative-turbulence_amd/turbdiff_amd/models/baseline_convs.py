"""The convolution layers of the reference's regression baselines on the HIP kernels.

``DilatedCNNBlock`` (turbdiff/models/dilresnet.py:22-44), and tfnet's ``conv`` / ``deconv`` layer factories
(turbdiff/models/tfnet.py:185-208), with the reference's parameter containers (``nn.Conv3d`` /
``nn.ConvTranspose3d`` / ``nn.BatchNorm3d`` in the same attribute paths, so ``state_dict``s are interchangeable)
and NDHWC activations ``(B, X, Y, Z, C)`` executed by ``ops.conv3d`` / ``ops.conv_transpose3d``.  These layers are
off the benchmark path: bf16 tensors run the matrix-core gather / transposed / weight-gradient kernels of
``csrc/tdx_convg_mfma.hip`` (unstaged 16-B fragment loads; 20x the vector-ALU kernels, 2-60x MIOpen on the same GPU),
fp32 tensors the vector-ALU family of ``csrc/tdx_convg.hip``; no fusion beyond the bias.  ``to_ndhwc`` / ``to_ncdhw`` convert at a
model boundary (``ops.to_nvc`` / ``ops.to_ncv``)."""

from __future__ import annotations

import itertools as it

import torch
import torch.nn.functional as F
from torch import nn

from .. import ops


class DilatedCNNBlock(nn.Module):
    """dilresnet.py:22-44: 3x3x3 replicate-padded convs with dilations d1..dn..d1, ReLU after each."""

    def __init__(self, dim: int, dilations: list[int]):
        super().__init__()
        self.dim, self.dilations = dim, dilations
        self.layers = nn.ModuleList([
            nn.Conv3d(dim, dim, kernel_size=3, dilation=d, padding=d, padding_mode="replicate")
            for d in it.chain(dilations, reversed(dilations[:-1]))])

    def forward(self, x):
        for layer in self.layers:
            d = layer.dilation[0]
            x = F.relu(ops.conv3d(x, layer.weight, layer.bias, dilation=d, padding=d, padding_mode="replicate"))
        return x


def _batch_norm_channels_last(bn: nn.BatchNorm3d, x: torch.Tensor) -> torch.Tensor:
    """BatchNorm3d's parameters / running statistics applied to an NDHWC tensor (statistics over all but the last axis)."""
    y = F.batch_norm(x.reshape(-1, x.shape[-1]).float(), bn.running_mean, bn.running_var, bn.weight, bn.bias, bn.training,
                     bn.momentum, bn.eps)
    return y.reshape(x.shape).to(x.dtype)


class ConvBlock(nn.Sequential):
    """tfnet.py:185-199 ``conv(...)``: Conv3d(k, stride, padding=(k-1)//2) -> BatchNorm3d -> LeakyReLU(0.1) -> Dropout;
    children 0..3 as in the reference's ``nn.Sequential``."""

    def __init__(self, input_channels, output_channels, kernel_size, stride, dropout_rate):
        super().__init__(nn.Conv3d(input_channels, output_channels, kernel_size=kernel_size, stride=stride,
                                   padding=(kernel_size - 1) // 2),
                         nn.BatchNorm3d(output_channels), nn.LeakyReLU(0.1, inplace=True), nn.Dropout(dropout_rate))

    def forward(self, x):
        c, bn, act, drop = self[0], self[1], self[2], self[3]
        h = ops.conv3d(x, c.weight, c.bias, stride=c.stride[0], padding=c.padding[0])
        return drop(F.leaky_relu(_batch_norm_channels_last(bn, h), act.negative_slope))


class DeconvBlock(nn.Sequential):
    """tfnet.py:201-208 ``deconv(...)``: ConvTranspose3d(4, stride 2, padding 1) -> LeakyReLU(0.1)."""

    def __init__(self, input_channels, output_channels):
        super().__init__(nn.ConvTranspose3d(input_channels, output_channels, kernel_size=4, stride=2, padding=1),
                         nn.LeakyReLU(0.1, inplace=True))

    def forward(self, x):
        c = self[0]
        return F.leaky_relu(ops.conv_transpose3d(x, c.weight, c.bias, stride=c.stride[0], padding=c.padding[0]), self[1].negative_slope)


def conv(input_channels, output_channels, kernel_size, stride, dropout_rate):
    return ConvBlock(input_channels, output_channels, kernel_size, stride, dropout_rate)


def deconv(input_channels, output_channels):
    return DeconvBlock(input_channels, output_channels)


def to_ndhwc(x: torch.Tensor, dtype: torch.dtype = torch.float32):
    return ops.to_nvc(x, dtype)


def to_ncdhw(x: torch.Tensor):
    return ops.to_ncv(x, torch.float32)
