"""Drop-in for the reference's ``models/stylegan2/op/conv2d_gradfix.py`` (interface :22-75).

The reference module wraps cuDNN's convolutions in a custom autograd function on torch 1.7 / 1.8 only; on every later torch it warns and calls
``torch.nn.functional`` (``could_use_op`` :78-92) — which is what runs under the reference's own pins (torch 2.0.1) and here.  So this file is
that interface on top of ``F.conv2d`` / ``F.conv_transpose2d``, without the warning.  It is NOT on the engine's hot path (the modulated
convolutions run on the HIP kernels through ``ops``): it exists so that ``from models.stylegan2.op import conv2d_gradfix`` — the training-only
callers ``criteria/adv_loss.py:4`` and the Discriminator — keeps resolving after ``e4s2024_amd.install()``.
"""
import contextlib

from torch.nn import functional as F

enabled = True
weight_gradients_disabled = False


@contextlib.contextmanager
def no_weight_gradients():
    """Reference :14-20: a flag the custom backward consulted; stock autograd has no such switch, so it only records the request."""
    global weight_gradients_disabled
    previous, weight_gradients_disabled = weight_gradients_disabled, True
    try:
        yield
    finally:
        weight_gradients_disabled = previous


def conv2d(input, weight, bias=None, stride=1, padding=0, dilation=1, groups=1):
    return F.conv2d(input, weight, bias, stride, padding, dilation, groups)


def conv_transpose2d(input, weight, bias=None, stride=1, padding=0, output_padding=0, groups=1, dilation=1):
    return F.conv_transpose2d(input, weight, bias, stride, padding, output_padding, groups, dilation)


def could_use_op(input):
    return False
