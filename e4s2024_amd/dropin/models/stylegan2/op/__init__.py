"""Drop-in for the reference's ``models/stylegan2/op`` (op/__init__.py:1-2).  Nothing is compiled at import: the two ops
call the ahead-of-time built libe4s_hip.so through ctypes."""
from .fused_act import FusedLeakyReLU, fused_leaky_relu
from .upfirdn2d import upfirdn2d
