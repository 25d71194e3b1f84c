"""Drop-in for models/stylegan2/op/upfirdn2d.py (reference :142-147)."""
from e4s2024_amd import ops


def upfirdn2d(input, kernel, up=1, down=1, pad=(0, 0)):
    return ops.upfirdn2d(input, kernel, up=up, down=down, pad=pad)
