"""Drop-in for models/stylegan2/op/fused_act.py (reference :72-85)."""
import torch
from torch import nn

from e4s2024_amd import ops


class FusedLeakyReLU(nn.Module):
    def __init__(self, channel, negative_slope=0.2, scale=2 ** 0.5):
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(channel))
        self.negative_slope = negative_slope
        self.scale = scale

    def forward(self, input):
        return fused_leaky_relu(input, self.bias, self.negative_slope, self.scale)


def fused_leaky_relu(input, bias, negative_slope=0.2, scale=2 ** 0.5):
    return ops.fused_leaky_relu(input, bias, negative_slope, scale)
