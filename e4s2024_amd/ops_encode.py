"""Rows a8 - a10 of the scope table: the regional-style encoder's and the face parser's operators (``csrc/conv.hip``, ``csrc/conv_mx3.hip``,
``csrc/winograd.hip``, ``csrc/norm.hip``, ``csrc/parser.hip``, ``csrc/stem7.hip``) — prepared convolution weights, the route choice of a 3 x 3 convolution,
instance-norm / SE-gate / shortcut fusions, pooling, resizing, argmax.  Reference: ``models/encoders/psp_encoders.py:319-401``, ``models/encoders/helpers.py:56-144``,
``swap_face_fine/face_parsing/model.py:20-260``, ``face_parsing_demo.py:15-200``.

Re-exported by ``ops``; the stage's switches (``ops.WINOGRAD``, ``ops.MX3``, ``ops.SE_GATE_IS_HALF`` ...) live in ``ops`` and are read there at call time.
"""
from __future__ import annotations

import math
import os
from typing import Optional

import torch

from . import ops
from ._lib import lib
from .ops import _Prepared, _c, _p, _stream, _timed, _volatile, mx_arith, mx_exact_active, mx_flags

# --------------------------------------------------------------------------- a8 / a9 (conv.hip, norm.hip, parser.hip)
class PreparedConv(_Prepared):
    """K-major copy of a plain conv weight ``[cout, cin, k, k]`` (optionally with an eval-mode BatchNorm2d folded in),
    rebuilt when a parameter or BN buffer changes version or storage.  ``get`` returns the prepared copy as an immutable record
    ``(wt, bias, shape)`` (attributes), which is what ``conv2d`` takes."""

    __slots__ = ("exact",)

    class Copy(tuple):
        __slots__ = ()
        wt = property(lambda self: self[0])
        bias = property(lambda self: self[1])
        shape = property(lambda self: self[2])
        kexp = property(lambda self: self[3] if len(self) > 3 else None)       # (f16x3) log2 of the weights' pre-scale

    def __init__(self, exact=False):
        """``exact=True`` pins this convolution to the exact-fp32 MFMA kernel whatever ``ops.CONV_MODE`` says (the face parser:
        its argmax must match the reference pixel for pixel, and split-bf16's ~2e-5 relative logit error flips near-ties);
        ``exact="sb3"`` asks for the three-way bf16 split (fp32-class error) where a split kernel exists, fp32 elsewhere; ``exact="f16x3"`` for the
        two-term f16 split (the same error class at half the MFMAs; its preparation reads the largest folded weight back to pick a power-of-two
        scale — one host sync per weight version, so not for weights prepared inside a graph capture)."""
        super().__init__()
        self.exact = exact

    def __reduce__(self):
        return (self.__class__, (self.exact,))

    def use_sb(self, cin: int, kh: int, kw: int) -> int:
        """Number of bf16 terms per operand: 2 (``wt = (whi, wlo)``) or 3 (``(w0, w1, w2)``) for 3x3 / 1x1 kernels with at least 16
        input channels, 0 = exact-fp32 kernel (always for the 3-channel stems: 7x7 ResNet stem, encoder input layer)."""
        if not (kh == kw and kh in (1, 3) and cin >= 16):
            return 0
        if self.exact == "sb3":
            return 3
        if self.exact == "f16x3":
            return 3 if mx_exact_active() else 4          # (the re-run of a pass whose f16 arithmetic overflowed: three-way bf16 split)
        return 2 if (ops.CONV_MODE == "sb" and not self.exact) else 0

    def get(self, weight: torch.Tensor, bn=None, conv_bias: Optional[torch.Tensor] = None):
        ts = [weight] + ([bn.weight, bn.bias, bn.running_mean, bn.running_var] if bn is not None else []) + ([conv_bias] if conv_bias is not None else [])
        key = tuple((t.data_ptr(), t._version) for t in ts) + (weight.device, ops.CONV_MODE, self.exact == "f16x3" and mx_exact_active())
        if any(_volatile(t) for t in ts):
            key = None
        hit = self._lookup(key)
        if hit is not None:
            return hit
        w = _c(weight.detach(), "weight")
        cout, cin, kh, kw = w.shape
        if ops.STEM7 and self.exact == "f16x3" and not mx_exact_active() and (cout, cin, kh, kw) == (64, 3, 7, 7) and conv_bias is None and w.is_cuda:
            # the parser's 7x7 stem on its own kernel (csrc/stem7.hip): K = (c, ky, kx) flattened; the two f16 terms of the BN-folded weight x 2^kexp, built here
            # (64 x 147 values, once per weight version; one host read of the largest folded weight like the f16x3 route below)
            with torch.no_grad():
                if bn is not None:
                    if bn.training:
                        raise RuntimeError("BatchNorm2d must be in eval mode to be folded into the convolution (the parser runs in eval mode)")
                    sc = bn.weight.detach().float() / torch.sqrt(bn.running_var.float() + float(bn.eps))
                    wf = w.float() * sc[:, None, None, None]
                    bias = (bn.bias.detach().float() - bn.running_mean.float() * sc).contiguous()
                else:
                    wf, bias = w.float(), None
                m = float(wf.abs().max().item())
                kexp = 10 - int(math.ceil(math.log2(m))) if m > 0 and math.isfinite(m) else 0
                kexp = max(-30, min(30, kexp))
                wk = torch.zeros((64, 160), dtype=torch.float32, device=w.device)
                wk[:, :147] = wf.reshape(64, 147) * float(2.0 ** kexp)
                hi = wk.half()
                lo = (wk - hi.float()).half()
                wt = torch.stack([hi, lo]).view(2, 64, 10, 2, 8).permute(0, 2, 3, 1, 4).contiguous().view(torch.int16)
            return self._publish(key, PreparedConv.Copy((wt, bias, (cout, cin, kh, kw), kexp, "stem7")))
        sb = self.use_sb(cin, kh, kw)
        if sb:
            shape = ((cin + 15) // 16, kh * kw, 2, cout, 8)
            wt = tuple(torch.empty(shape, dtype=torch.int16, device=w.device) for _ in range(2 if sb == 4 else sb))
        else:
            wt = torch.empty((cin, kh * kw, cout), dtype=torch.float32, device=w.device)
        bias = torch.empty((cout,), dtype=torch.float32, device=w.device) if (bn is not None or conv_bias is not None) else None
        if bn is not None:
            if bn.training:
                raise RuntimeError("BatchNorm2d must be in eval mode to be folded into the convolution (the parser runs in eval mode)")
            g, be, mu, var, eps = _c(bn.weight.detach(), "bn.weight"), _c(bn.bias.detach(), "bn.bias"), _c(bn.running_mean, "bn.running_mean"), \
                _c(bn.running_var, "bn.running_var"), float(bn.eps)
        else:
            g = be = mu = var = None
            eps = 0.0
        cb = _c(conv_bias.detach(), "conv bias") if conv_bias is not None else None
        kexp = 0
        if sb == 4:
            # power-of-two pre-scale: the largest folded weight lands in (2^9, 2^10], so that every weight's second f16 term stays normal
            with torch.no_grad():
                wmax = w.abs().flatten(1).amax(1)
                if bn is not None:
                    wmax = wmax * (g / torch.sqrt(var + eps)).abs()
                m = float(wmax.max().item())
            kexp = 10 - int(math.ceil(math.log2(m))) if m > 0 and math.isfinite(m) else 0
            kexp = max(-30, min(30, kexp))
            lib().call("e4s_conv_prep_weights_f16x3", _p(wt[0]), _p(wt[1]), _p(bias), _p(w), _p(g), _p(be), _p(mu), _p(var), eps, _p(cb), cout, cin, kh, kw,
                       kexp, _stream())
        elif sb == 3:
            lib().call("e4s_conv_prep_weights_sb3", _p(wt[0]), _p(wt[1]), _p(wt[2]), _p(bias), _p(w), _p(g), _p(be), _p(mu), _p(var), eps, _p(cb),
                       cout, cin, kh, kw, _stream())
        elif sb:
            lib().call("e4s_conv_prep_weights_sb", _p(wt[0]), _p(wt[1]), _p(bias), _p(w), _p(g), _p(be), _p(mu), _p(var), eps, _p(cb), cout,
                       cin, kh, kw, _stream())
        else:
            lib().call("e4s_conv_prep_weights", _p(wt), _p(bias), _p(w), _p(g), _p(be), _p(mu), _p(var), eps, _p(cb), cout, cin, kh, kw,
                       _stream())
        return self._publish(key, PreparedConv.Copy((wt, bias, (cout, cin, kh, kw), kexp, "f16x3" if sb == 4 else "")))




def _is_f16x3(prepared) -> bool:
    return len(prepared) > 4 and prepared[4] == "f16x3"


def conv2d(x: torch.Tensor, prepared: PreparedConv, stride: int = 1, pad: int = 0, *, x1: Optional[torch.Tensor] = None, in_norm=None,
           prelu: Optional[torch.Tensor] = None, relu: bool = False, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``act(conv2d(cat(x, x1), W) + bias + residual)``; ``in_norm=(mean, rstd)`` applies InstanceNorm to the input on load."""
    x = _c(x, "input")
    cout, cin, kh, kw = prepared.shape
    bs, c0, h, w = x.shape
    if kh != kw:
        raise NotImplementedError("square kernels only")
    if x1 is not None:
        x1 = _c(x1, "input (second half)")
        if x1.shape[0] != bs or tuple(x1.shape[2:]) != (h, w):
            raise ValueError("concatenated inputs must share batch and spatial size")
    if c0 + (0 if x1 is None else x1.shape[1]) != cin:
        raise ValueError(f"conv expects {cin} input channels, got {c0 + (0 if x1 is None else x1.shape[1])}")
    ho, wo = (h + 2 * pad - kh) // stride + 1, (w + 2 * pad - kh) // stride + 1
    out = torch.empty((bs, cout, ho, wo), dtype=torch.float32, device=x.device)
    mean = rstd = None
    if in_norm is not None:
        mean, rstd = _c(in_norm[0], "in_mean"), _c(in_norm[1], "in_rstd")
    act = 2 if prelu is not None else (1 if relu else 0)
    res = None
    if residual is not None:
        res = _c(residual, "residual")
        if tuple(res.shape) != tuple(out.shape):
            raise ValueError(f"residual shape {tuple(res.shape)} != output {tuple(out.shape)}")
    if len(prepared) > 4 and prepared[4] == "stem7":
        if (stride, pad) != (2, 3) or x1 is not None or in_norm is not None or prelu is not None or residual is not None:
            raise ValueError("this weight copy is the parser stem's (7x7, stride 2, pad 3, ReLU or nothing)")
        ev = _timed("conv7x7s2_stem_f16x3")
        lib().call("e4s_conv7x7s2_stem_f16x3", _p(out), _p(x), _p(prepared.wt), _p(prepared.bias), bs, h, w, 1 if relu else 0, prepared.kexp, _stream())
        if ev is not None:
            ev.record()
        return out
    sb = isinstance(prepared.wt, tuple)
    ev = _timed(f"conv2d_{'sb_' if sb else ''}kernel<{kh},{stride}>", f"{cin}->{cout} @{h}")
    pr = _p(_c(prelu.detach(), "prelu")) if prelu is not None else None
    if sb and _is_f16x3(prepared):
        lib().call("e4s_conv2d_f16x3", _p(out), _p(x), _p(x1), c0, _p(prepared.wt[0]), _p(prepared.wt[1]), _p(prepared.bias), _p(mean), _p(rstd),
                   pr, _p(res), act, bs, cin, cout, h, w, kh, stride, pad, prepared.kexp, _p(mx_flags(x.device)), _stream())
    elif sb and len(prepared.wt) == 3:
        lib().call("e4s_conv2d_sb3", _p(out), _p(x), _p(x1), c0, _p(prepared.wt[0]), _p(prepared.wt[1]), _p(prepared.wt[2]), _p(prepared.bias),
                   _p(mean), _p(rstd), pr, _p(res), act, bs, cin, cout, h, w, kh, stride, pad, _stream())
    elif sb:
        lib().call("e4s_conv2d_sb", _p(out), _p(x), _p(x1), c0, _p(prepared.wt[0]), _p(prepared.wt[1]), _p(prepared.bias), _p(mean), _p(rstd),
                   pr, _p(res), act, bs, cin, cout, h, w, kh, stride, pad, _stream())
    else:
        lib().call("e4s_conv2d", _p(out), _p(x), _p(x1), c0, _p(prepared.wt), _p(prepared.bias), _p(mean), _p(rstd), pr, _p(res), act, bs, cin,
                   cout, h, w, kh, stride, pad, _stream())
    if ev is not None:
        ev.record()
    return out


# ---- Winograd F(2x2, 3x3) route of the encoder's stride-1 3x3 convolutions (csrc/winograd.hip + the batched split-bf16 GEMM)


class PreparedWinograd(_Prepared):
    """``U [16, cout, cin] = G g G^T`` of a 3x3 conv weight, rebuilt when the parameter changes version or storage (``e4s_wino_weight``)."""

    __slots__ = ()

    def get(self, weight: torch.Tensor) -> torch.Tensor:
        key = None if _volatile(weight) else ((weight.data_ptr(), weight._version), weight.device)
        hit = self._lookup(key)
        if hit is not None:
            return hit[0]
        w = _c(weight.detach(), "weight")
        cout, cin, kh, kw = w.shape
        if (kh, kw) != (3, 3):
            raise ValueError("PreparedWinograd: 3x3 kernels only")
        U = torch.empty((16, cout, cin), dtype=torch.float32, device=w.device)
        lib().call("e4s_wino_weight", _p(U), _p(w), cout, cin, _stream())
        return self._publish(key, (U,))[0]


def winograd_route(x: torch.Tensor, cin: int, stride: int):
    """Which route a 3x3, pad-1 convolution of ``x`` takes: ``"f32"`` (Winograd F(2x2, 3x3) on the general split-bf16 GEMM) or ``None`` (the direct /
    DMA-fed kernels).  Stride 1, even maps, inference only; at least ``ops.WINOGRAD_MIN_CIN`` channels and between ``ops.WINOGRAD_MIN_TILES`` and
    ``ops.WINOGRAD_MAX_TILES`` 2 x 2 output tiles (above that the two transforms cost more than the GEMMs save once the direct kernel fills the chip:
    256 -> 256 @64^2 at 16 faces 0.31 against 0.28 ms).  ``ops.ENC_ROUTE_BY_IMAGE``: the tile count is taken PER IMAGE, so a face's style vectors do
    not depend on how many faces share the launch.  (Round 2-3 also carried a route with operands split to bf16 by their producers; it gave rare wrong
    values beside a second stream, was never root-caused and stayed off — removed in round 4.)"""
    bs, _, h, w = x.shape
    if not (ops.WINOGRAD and stride == 1 and x.is_cuda and not torch.is_grad_enabled() and cin >= ops.WINOGRAD_MIN_CIN and h % 2 == 0 and w % 2 == 0):
        return None
    tiles = (1 if ops.ENC_ROUTE_BY_IMAGE else bs) * (h // 2) * (w // 2)
    if tiles < ops.WINOGRAD_MIN_TILES or tiles > ops.WINOGRAD_MAX_TILES:
        return None
    return "f32"






def mx4_eligible(cin: int, cout: int, h: int, w: int, bs: int) -> bool:
    """Does a masked up layer ``[bs, cin, h, w] -> [bs, cout, 2h, 2w]`` (one ``mx_eligible`` accepts, f16 + fp6 arithmetic in force) try the four-parity kernel?
    Its workgroup is 64 output channels x (32 x 8) positions x 4 parities — twice the composed kernel's work — so the launch must still fill the chip (one workgroup
    per CU), and the layer must not be one the region-uniform block path takes (``ops.UP_BLOCKS_MIN_WIDTH``)."""
    if not ops.UP_MX4 or cin % 16 or cout % 128 or w < 32 or (ops.UP_BLOCKS and w >= ops.UP_BLOCKS_MIN_WIDTH and cout >= 128):
        return False
    return (-(-w // 32)) * (-(-h // 8)) * (-(-cout // 64)) * bs >= 256


def mx_conv_eligible(x: torch.Tensor, cout: int) -> bool:
    """Does a stride-1 3x3 convolution of ``x`` run on the DMA-fed kernel's plain-convolution mode?  (inference, the split arithmetic in force,
    16-channel chunks, >= 128 output channels, maps at least 32 wide.)  Layers below ``ops.WINOGRAD_MIN_CIN`` input channels decide from the image alone
    — at least ``ops.MX_CONV_MIN_WORKGROUPS_PER_IMAGE`` 128 co x (32 x 8) px tiles per image — so that a face's style vectors do not depend on how many
    faces share the batch; the 256- / 512-channel layers, whose Winograd route already depends on the launch size, take it when the whole launch
    has ``ops.MX_CONV_MIN_WORKGROUPS`` tiles (the full swap's 16 images; smaller batches keep Winograd / the direct kernel)."""
    bs, cin, h, w = x.shape
    if mx_arith() is None or ops.CONV_MODE != "sb" or torch.is_grad_enabled() or not x.is_cuda:
        return False
    if cin % 16 or cout < 128 or w < 32:
        return False
    per_image = (-(-w // 32)) * (-(-h // 8)) * (-(-cout // 128))
    if cin < ops.WINOGRAD_MIN_CIN:
        return per_image >= ops.MX_CONV_MIN_WORKGROUPS_PER_IMAGE
    if ops.ENC_ROUTE_BY_IMAGE:
        return per_image >= ops.MX_CONV_MIN_WORKGROUPS_PER_IMAGE
    return bs * per_image >= ops.MX_CONV_MIN_WORKGROUPS


class MxOperandMap:
    """The hand-over of ``conv3x3_mx(out_prep=True)``: an activation map ``[bs, c, h, w]`` stored as the f16 + fp6 OPERANDS of the consuming two-phase convolution
    (per image and 32-channel block ``116 h w`` bytes: f16 part, two fp6 code sets, block scales — ``e4s_conv3x3_mx3_ex``, layout bit 4), in plain or phase-plane pixel
    order.  Only ``conv3x3_mx`` / ``conv3x3_s2_mx`` (through ``conv3x3_s1`` / ``conv3x3_s2``) read it; ``data`` is ``float32 [bs, c / 32, 29 h w]`` raw storage."""
    __slots__ = ("data", "bs", "c", "h", "w", "phased")

    def __init__(self, data, bs, c, h, w, phased):
        self.data, self.bs, self.c, self.h, self.w, self.phased = data, bs, c, h, w, bool(phased)

    shape = property(lambda self: (self.bs, self.c, self.h, self.w))
    device = property(lambda self: self.data.device)
    is_cuda = property(lambda self: self.data.is_cuda)

    def dim(self):
        return 4


def conv3x3_mx(x, wmx: torch.Tensor, arith: int, cout: int, *, in_norm=None, prelu: Optional[torch.Tensor] = None,
               out_phased: bool = False, out_c4: bool = False, out_prep: bool = False):
    """``PReLU(conv3x3(norm(x), W))``, stride 1, pad 1, on ``e4s_conv3x3_mx`` / ``e4s_conv3x3_mx3`` (``arith`` 3) (``wmx`` from ``PreparedMx.get`` of the
    plain weight with the same ``arith``).  Hand-over layouts of the two-phase kernel (``arith`` 3), which only another ``conv3x3_mx`` / ``conv3x3_s2_mx`` reads:
    ``out_phased`` (even maps): the result's MEMORY is phase planes — ``[bs, cout, 2, 2, h / 2, w / 2]``, plane ``(py, px)`` = ``result[..., py::2, px::2]``;
    ``out_c4``: channel-blocked — ``[bs, cout / 4, h, w, 4]`` (with ``out_phased``: ``[bs, cout / 4, 2, 2, h / 2, w / 2, 4]``), a pixel's four channels one
    16-byte element.  The returned tensor has that shape; a 5-D INPUT is such a channel-blocked map.  ``out_prep``: the result as the consumer's OPERANDS
    (``MxOperandMap``; with ``out_phased`` in phase-plane pixel order), ``cout % 32 == 0``; an ``MxOperandMap`` INPUT takes no ``in_norm``.  Same values in every layout."""
    in_prep = isinstance(x, MxOperandMap)
    if in_prep:
        if arith != 3 or in_norm is not None or x.phased:
            raise ValueError("conv3x3_mx: a prepared-operand input goes to the two-phase kernel, un-normalised, in plain pixel order")
        bs, cin, h, w = x.shape
        x, in_c4 = _c(x.data, "input"), False
    else:
        x = _c(x, "input")
        in_c4 = x.dim() == 5
    if in_prep:
        pass
    elif in_c4:
        if x.shape[4] != 4:
            raise ValueError("conv3x3_mx: a channel-blocked input is [bs, cin / 4, h, w, 4]")
        bs, cin, h, w = x.shape[0], 4 * x.shape[1], x.shape[2], x.shape[3]
    else:
        bs, cin, h, w = x.shape
    mean = rstd = None
    if in_norm is not None:
        mean, rstd = _c(in_norm[0], "in_mean"), _c(in_norm[1], "in_rstd")
    pr = _p(_c(prelu.detach(), "prelu")) if prelu is not None else None
    if out_phased or out_c4 or in_c4 or in_prep or out_prep:
        if arith != 3 or (out_phased and (h % 2 or w % 2)) or (out_c4 and cout % 4) or (out_prep and (cout % 32 or (h * w) % 4 or out_c4)):
            raise ValueError("conv3x3_mx: hand-over layouts need the two-phase kernel (arith 3); phase planes an even map, channel blocks cout % 4 == 0, "
                             "prepared operands cout % 32 == 0 and h w % 4 == 0")
        if out_prep:
            out = torch.empty((bs, cout // 32, 29 * h * w), dtype=torch.float32, device=x.device)
        else:
            shape = (bs, cout // 4 if out_c4 else cout) + ((2, 2, h // 2, w // 2) if out_phased else (h, w)) + ((4,) if out_c4 else ())
            out = torch.empty(shape, dtype=torch.float32, device=x.device)
        ev = _timed("conv3x3_mx<3>", f"{cin}->{cout} @{h}")
        lib().call("e4s_conv3x3_mx3_ex", _p(out), _p(x), _p(wmx), _p(mx_flags(x.device)), _p(mean), _p(rstd), pr, bs, cin, cout, h, w,
                   4 if in_prep else 2 if in_c4 else 0, (1 if out_phased else 0) | (2 if out_c4 else 0) | (4 if out_prep else 0), _stream())
        if ev is not None:
            ev.record()
        return MxOperandMap(out, bs, cout, h, w, out_phased) if out_prep else out
    out = torch.empty((bs, cout, h, w), dtype=torch.float32, device=x.device)
    ev = _timed(f"conv3x3_mx<{arith}>", f"{cin}->{cout} @{h}")
    if arith == 3:
        lib().call("e4s_conv3x3_mx3", _p(out), _p(x), _p(wmx), _p(mx_flags(x.device)), _p(mean), _p(rstd), pr, bs, cin, cout, h, w, _stream())
    else:
        lib().call("e4s_conv3x3_mx", _p(out), _p(x), _p(wmx), arith, _p(mx_flags(x.device)) if arith else None, _p(mean), _p(rstd), pr, bs, cin, cout, h, w, _stream())
    if ev is not None:
        ev.record()
    return out


def conv3x3_s2_mx(x: torch.Tensor, wmx: torch.Tensor, cout: int, *, in_norm=None, prelu: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``PReLU(conv3x3(norm(x), W, stride 2, pad 1))`` on ``e4s_conv3x3_s2_mx3`` (f16 + 2 x MX fp6; ``wmx`` from ``PreparedMx.get(weight, None, False, 5)``);
    the input's height and width must be even, ``cin % 32 == 0``, ``cin <= 512``.  A 6-D input ``[bs, cin, 2, 2, h / 2, w / 2]`` is the phase-plane
    hand-over of ``conv3x3_mx(out_phased=True)``, a 7-D one ``[bs, cin / 4, 2, 2, h / 2, w / 2, 4]`` that of ``conv3x3_mx(out_phased=True, out_c4=True)``, an
    ``MxOperandMap`` that of ``conv3x3_mx(out_prep=True)`` (no ``in_norm`` then)."""
    in_prep = isinstance(x, MxOperandMap)
    if in_prep:
        if in_norm is not None:
            raise ValueError("conv3x3_s2_mx: a prepared-operand input is not normalised")
        bs, cin, h, w = x.shape
        in_phased, in_c4, x = x.phased, False, _c(x.data, "input")
    else:
        x = _c(x, "input")
        in_phased, in_c4 = x.dim() in (6, 7), x.dim() == 7
    if in_prep:
        pass
    elif in_phased:
        if x.shape[2] != 2 or x.shape[3] != 2 or (in_c4 and x.shape[6] != 4):
            raise ValueError("conv3x3_s2_mx: a phase-plane input is [bs, cin, 2, 2, h / 2, w / 2] or [bs, cin / 4, 2, 2, h / 2, w / 2, 4]")
        bs, cin, h, w = x.shape[0], x.shape[1] * (4 if in_c4 else 1), 2 * x.shape[4], 2 * x.shape[5]
    else:
        bs, cin, h, w = x.shape
    if h % 2 or w % 2:
        raise ValueError("conv3x3_s2_mx: the input height and width must be even")
    out = torch.empty((bs, cout, h // 2, w // 2), dtype=torch.float32, device=x.device)
    mean = rstd = None
    if in_norm is not None:
        mean, rstd = _c(in_norm[0], "in_mean"), _c(in_norm[1], "in_rstd")
    ev = _timed("conv3x3_s2_mx<3>", f"{cin}->{cout} @{h}")
    pr = _p(_c(prelu.detach(), "prelu")) if prelu is not None else None
    lib().call("e4s_conv3x3_s2_mx3", _p(out), _p(x), _p(wmx), _p(mx_flags(x.device)), _p(mean), _p(rstd), pr, bs, cin, cout, h, w,
               (1 if in_phased else 0) | (2 if in_c4 else 0) | (4 if in_prep else 0), _stream())
    if ev is not None:
        ev.record()
    return out


def conv3x3_s2_takes_mx(bs: int, cin: int, cout: int, h: int, w: int, device) -> bool:
    """Does a stride-2 3x3 convolution of a ``[bs, cin, h, w]`` map run on ``e4s_conv3x3_s2_mx3``?  (``mx_conv_eligible`` of the output-sized launch.)"""
    if not (ops.S2_MX3 and ops.MX3 and mx_arith() == 1 and cin % 32 == 0 and cin <= 512 and h % 2 == 0 and w % 2 == 0):
        return False
    return mx_conv_eligible(_ShapeOnly(bs, cin, h // 2, w // 2, device), cout)


class _ShapeOnly:
    """What ``mx_conv_eligible`` looks at of its input (shape, device kind) for a map that does not exist yet."""
    __slots__ = ("shape", "is_cuda")

    def __init__(self, bs, c, h, w, device):
        self.shape = (bs, c, h, w)
        self.is_cuda = torch.device(device).type == "cuda"


def conv3x3_s2(x: torch.Tensor, weight: torch.Tensor, caches) -> torch.Tensor:
    """A stride-2, pad-1 3x3 convolution: the DMA-fed f16 + fp6 kernel where it fits and fills the chip (``conv3x3_s2_takes_mx``), else the direct kernel.
    ``caches = (PreparedConv, PreparedWinograd, PreparedMx)`` of the layer.  ``x`` may be the phase-plane hand-over of ``conv3x3_s1(out_phased=True)``."""
    if isinstance(x, MxOperandMap) or x.dim() in (6, 7):
        return conv3x3_s2_mx(x, caches[2].get(weight, None, False, 5), weight.shape[0])
    bs, cin, h, w = x.shape
    if len(caches) > 2 and conv3x3_s2_takes_mx(bs, cin, weight.shape[0], h, w, x.device):
        return conv3x3_s2_mx(x, caches[2].get(weight, None, False, 5), weight.shape[0])
    return conv2d(x, caches[0].get(weight), 2, 1)


def conv3x3_s1_takes_mx3(x: torch.Tensor, cout: int) -> bool:
    """``conv3x3_s1`` runs this layer on the two-phase kernel (``e4s_conv3x3_mx3``)."""
    return (winograd_route(x, x.shape[1], 1) != "f32" and mx_conv_eligible(x, cout) and mx_arith() == 1 and ops.MX3
            and x.shape[1] % 32 == 0 and x.shape[1] <= 512)


def conv3x3_s1_c4_pair(x: torch.Tensor, depth: int, cout2: int, stride2: int) -> bool:
    """Do BOTH 3x3 convolutions of an IR-SE unit — ``x -> depth`` at stride 1, ``depth -> cout2`` at ``stride2`` — run on the two-phase kernel, so that the map between
    them can be handed over channel-blocked (``ops.ENC_C4_LINK``) or as prepared operands (``ops.ENC_PREP_LINK``: ``depth % 32 == 0``, ``h w % 4 == 0`` on top)?"""
    if not ((ops.ENC_C4_LINK or ops.ENC_PREP_LINK) and x.dim() == 4 and depth % 4 == 0 and conv3x3_s1_takes_mx3(x, depth)):
        return False
    bs, _, h, w = x.shape
    if stride2 == 2:
        return h % 2 == 0 and w % 2 == 0 and conv3x3_s2_takes_mx(bs, depth, cout2, h, w, x.device)
    return stride2 == 1 and conv3x3_s1_takes_mx3(_ShapeOnly(bs, depth, h, w, x.device), cout2)


def conv3x3_s1(x, weight: torch.Tensor, caches, *, in_norm=None, prelu: Optional[torch.Tensor] = None, out_phased: bool = False,
               out_c4: bool = False, out_prep: bool = False):
    """A stride-1, pad-1 3x3 convolution by whichever route fits the launch: Winograd (``winograd_route``: small batches), the DMA-fed kernel
    (``mx_conv_eligible``: launches that fill the chip) or the direct kernel; ``caches = (PreparedConv, PreparedWinograd, PreparedMx)`` of the layer.
    ``out_phased``: see ``conv3x3_mx`` — the caller has checked ``conv3x3_s1_takes_mx3``."""
    if isinstance(x, MxOperandMap) or x.dim() == 5:          # the hand-over of a conv3x3_s1(out_prep / out_c4 = True): the producer checked conv3x3_s1_c4_pair
        return conv3x3_mx(x, caches[2].get(weight, None, False, 3), 3, weight.shape[0], in_norm=in_norm, prelu=prelu, out_phased=out_phased, out_c4=out_c4, out_prep=out_prep)
    if out_phased or out_c4 or out_prep:
        if not (len(caches) > 2 and conv3x3_s1_takes_mx3(x, weight.shape[0])):
            raise RuntimeError("conv3x3_s1: a hand-over layout on a layer that does not run on the two-phase kernel")
        return conv3x3_mx(x, caches[2].get(weight, None, False, 3), 3, weight.shape[0], in_norm=in_norm, prelu=prelu, out_phased=out_phased, out_c4=out_c4, out_prep=out_prep)
    route = winograd_route(x, x.shape[1], 1)
    if route == "f32":
        return conv2d_winograd(x, caches[1].get(weight), in_norm=in_norm, prelu=prelu)
    if len(caches) > 2 and mx_conv_eligible(x, weight.shape[0]):
        arith = mx_arith()
        if arith == 1 and ops.MX3 and x.shape[1] % 32 == 0 and x.shape[1] <= 512:
            arith = 3             # same arithmetic, the two-phase kernel (csrc/conv_mx3.hip)
        return conv3x3_mx(x, caches[2].get(weight, None, False, arith), arith, weight.shape[0], in_norm=in_norm, prelu=prelu)
    return conv2d(x, caches[0].get(weight), 1, 1, in_norm=in_norm, prelu=prelu)


def conv2d_winograd(x: torch.Tensor, U: torch.Tensor, *, in_norm=None, prelu: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``act(conv2d(norm(x), W, stride 1, pad 1))`` for a 3x3 kernel through Winograd F(2x2, 3x3): input transform (with the InstanceNorm
    of ``in_norm=(mean, rstd)`` applied on load), 16 GEMMs ``U_k [cout, cin] @ V_k [cin, tiles]`` on the split-bf16 MFMA GEMM, output
    transform with the PReLU.  2.25x fewer multiplications than the direct kernel; same results to ~1e-5 relative."""
    x = _c(x, "input")
    bs, cin, h, w = x.shape
    if U.dim() != 3 or U.shape[0] != 16 or U.shape[2] != cin:
        raise ValueError(f"conv2d_winograd: U {tuple(U.shape)} does not fit {cin} input channels")
    cout = U.shape[1]
    T = bs * (h // 2) * (w // 2)
    mean = rstd = None
    if in_norm is not None:
        mean, rstd = _c(in_norm[0], "in_mean"), _c(in_norm[1], "in_rstd")
    V = torch.empty((16, cin, T), dtype=torch.float32, device=x.device)
    ev = _timed("conv2d_winograd<3,1>")
    lib().call("e4s_wino_input", _p(V), _p(x), _p(mean), _p(rstd), bs, cin, h, w, _stream())
    M = ops.gemm_sb(U, V, True, False, split_k=False)                                    # [16, cout, T]; no K split: a face's result does not depend on the batch
    del V
    out = torch.empty((bs, cout, h, w), dtype=torch.float32, device=x.device)
    lib().call("e4s_wino_output", _p(out), _p(M), _p(_c(prelu.detach(), "prelu")) if prelu is not None else None, bs, cout, h, w, _stream())
    if ev is not None:
        ev.record()
    return out


def plane_stats(x: torch.Tensor, eps: Optional[float] = None, want_nmean: bool = False):
    """Per-(b, c) mean [bs, C] (``eps=None``: mean only = global average pooling), or (mean, rstd[, nmean])."""
    x = _c(x, "input")
    bs, C = x.shape[:2]
    hw = x[0, 0].numel()
    mean = torch.empty((bs, C), dtype=torch.float32, device=x.device)
    if eps is None:
        lib().call("e4s_plane_stats", _p(mean), None, None, _p(x), bs * C, hw, 0.0, _stream())
        return mean
    rstd = torch.empty_like(mean)
    nmean = torch.empty_like(mean) if want_nmean else None
    lib().call("e4s_plane_stats", _p(mean), _p(rstd), _p(nmean), _p(x), bs * C, hw, float(eps), _stream())
    return (mean, rstd, nmean) if want_nmean else (mean, rstd)


ACT_NONE, ACT_RELU, ACT_SIGMOID = 0, 1, 3


def vec_fc(x: torch.Tensor, weight: torch.Tensor, bn=None, act: int = ACT_NONE) -> torch.Tensor:
    """``act(bn(x @ W^T))`` for ``x [bs, cin]`` and a 1x1 conv weight ``[cout, cin, 1, 1]`` (or ``[cout, cin]``)."""
    x = _c(x, "input")
    w = _c(weight.detach(), "weight")
    cout, cin = w.shape[0], w.shape[1]
    bs = x.shape[0]
    y = torch.empty((bs, cout), dtype=torch.float32, device=x.device)
    if bn is not None:
        if bn.training:
            raise RuntimeError("BatchNorm2d must be in eval mode")
        g, be, mu, var, eps = _c(bn.weight.detach(), "bn.weight"), _c(bn.bias.detach(), "bn.bias"), _c(bn.running_mean, "bn.running_mean"), \
            _c(bn.running_var, "bn.running_var"), float(bn.eps)
    else:
        g = be = mu = var = None
        eps = 0.0
    lib().call("e4s_vec_fc", _p(y), _p(x), _p(w), _p(g), _p(be), _p(mu), _p(var), eps, act, bs, cin, cout, _stream())
    return y


def se_gate(pooled: torch.Tensor, fc1_weight: torch.Tensor, fc2_weight: torch.Tensor) -> torch.Tensor:
    """``sigmoid(fc2 . relu(fc1 . pooled))`` for ``pooled [bs, C]`` and the two bias-free 1x1 conv weights of an SEModule, one launch
    (``e4s_se_gate``); value for value the two ``vec_fc`` calls."""
    x = _c(pooled, "pooled")
    w1, w2 = _c(fc1_weight.detach(), "fc1.weight"), _c(fc2_weight.detach(), "fc2.weight")
    bs, C = x.shape
    H = w1.shape[0]
    if w1.numel() != H * C or w2.numel() != C * H or w2.shape[0] != C or H > 64:
        raise ValueError(f"se_gate: fc1 {tuple(w1.shape)} / fc2 {tuple(w2.shape)} do not fit {C} channels (hidden width <= 64)")
    gate = torch.empty((bs, C), dtype=torch.float32, device=x.device)
    lib().call("e4s_se_gate", _p(gate), _p(x), _p(w1), _p(w2), bs, C, H, _stream())
    return gate


_half_gates = {}


def half_gate(bs: int, C: int, device) -> torch.Tensor:
    """``[bs, C]`` filled with 0.5, cached per shape and device for the life of the process (a few KB each; never evicted: a captured hipGraph may have the
    pointer baked in).  The fill runs on the stream that first asks; any other stream waits for its event before the first use (as ``_Prepared._lookup``
    does for the weight copies).  Inside a stream capture an uncached shape gets a fresh tensor that is not kept."""
    key = (torch.device(device), bs, C)
    ent = _half_gates.get(key)
    if ent is None:
        t = torch.full((bs, C), 0.5, dtype=torch.float32, device=device)
        if torch.cuda.is_current_stream_capturing():
            return t
        ev = torch.cuda.Event()
        ev.record()
        ent = _half_gates[key] = [t, ev, {torch.cuda.current_stream().cuda_stream}]
        return t
    t, ev, seen = ent
    sid = torch.cuda.current_stream().cuda_stream
    if sid not in seen:
        if not torch.cuda.is_current_stream_capturing():
            torch.cuda.current_stream().wait_event(ev)
            seen.add(sid)
        else:
            torch.cuda.current_stream().wait_event(ev)
    return t




def norm_gate_add(x, mean=None, rstd=None, gate=None, shortcut=None, sc_stats=None, sc_stride: int = 1, prelu=None, stats_eps: Optional[float] = None,
                  self_eps: Optional[float] = None):
    """``prelu(((x - mean) * rstd) * gate + shortcut')``.  With ``stats_eps`` the InstanceNorm statistics of the RESULT come back as well:
    ``(out, mean_out, rstd_out)`` — from the same launch for planes of up to 16384 pixels, from ``plane_stats`` otherwise.  ``self_eps`` (instead of ``mean`` / ``rstd``,
    with ``stats_eps``): the statistics of ``x`` itself are computed in that launch too (``e4s_norm_self_gate_add_stats``; ``plane_stats`` first where the plane does not fit)."""
    x = _c(x, "input")
    bs, C, h, w = x.shape
    if self_eps is not None:
        if mean is not None or rstd is not None or stats_eps is None:
            raise ValueError("norm_gate_add: self_eps replaces mean / rstd and goes with stats_eps")
        if not ((h * w) % 4 == 0 and (h * w <= ops.NGA_STATS_MAX_PIXELS or (h * w <= 4 * ops.NGA_STATS_MAX_PIXELS and shortcut is None))):
            mean, rstd = plane_stats(x, self_eps)
            self_eps = None
    out = torch.empty_like(x)
    sc = scm = scr = None
    if shortcut is not None:
        sc = _c(shortcut, "shortcut")
        if tuple(sc.shape) != (bs, C, h * sc_stride, w * sc_stride):
            raise ValueError(f"shortcut shape {tuple(sc.shape)} != {(bs, C, h * sc_stride, w * sc_stride)}")
        if sc_stats is not None:
            scm, scr = _c(sc_stats[0], "sc_mean"), _c(sc_stats[1], "sc_rstd")
    pr = _p(_c(prelu.detach(), "prelu")) if prelu is not None else None
    if self_eps is not None:
        om = torch.empty((bs, C), dtype=torch.float32, device=x.device)
        orr = torch.empty_like(om)
        lib().call("e4s_norm_self_gate_add_stats", _p(out), _p(om), _p(orr), _p(x), float(self_eps), _p(gate), _p(sc), _p(scm), _p(scr), sc_stride, pr,
                   bs, C, h, w, float(stats_eps), _stream())
        return out, om, orr
    if stats_eps is not None and (h * w) % 4 == 0 and h * w <= ops.NGA_STATS_MAX_PIXELS:
        om = torch.empty((bs, C), dtype=torch.float32, device=x.device)
        orr = torch.empty_like(om)
        lib().call("e4s_norm_gate_add_stats", _p(out), _p(om), _p(orr), _p(x), _p(mean), _p(rstd), _p(gate), _p(sc), _p(scm), _p(scr), sc_stride, pr,
                   bs, C, h, w, float(stats_eps), _stream())
        return out, om, orr
    lib().call("e4s_norm_gate_add", _p(out), _p(x), _p(mean), _p(rstd), _p(gate), _p(sc), _p(scm), _p(scr), sc_stride, pr, bs, C, h, w, _stream())
    if stats_eps is not None:
        return (out,) + tuple(plane_stats(out, stats_eps))
    return out


def masked_avg_pool(feats: torch.Tensor, labels: torch.Tensor, nreg: int) -> torch.Tensor:
    feats = _c(feats, "features")
    labels = _c(labels, "labels", torch.uint8)
    bs, C, h, w = feats.shape
    out = torch.empty((bs, nreg, C), dtype=torch.float32, device=feats.device)
    lib().call("e4s_masked_avg_pool", _p(out), _p(feats), _p(labels), labels.shape[1], labels.shape[2], bs, C, h, w, nreg, _stream())
    return out


def _out_like(out: Optional[torch.Tensor], shape, device, name: str) -> torch.Tensor:
    """``out=`` of the resize ops: a contiguous float32 CUDA tensor of exactly ``shape`` (e.g. one half of a batch buffer), or a new one."""
    if out is None:
        return torch.empty(shape, dtype=torch.float32, device=device)
    if not (isinstance(out, torch.Tensor) and out.is_cuda and out.dtype == torch.float32 and out.is_contiguous() and tuple(out.shape) == tuple(shape)):
        raise ValueError(f"{name}: out= must be a contiguous float32 CUDA tensor of shape {tuple(shape)}")
    return out


def bilinear_resize(x: torch.Tensor, size, align_corners: bool = False, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    x = _c(x, "input")
    bs, C, h, w = x.shape
    out = _out_like(out, (bs, C, size[0], size[1]), x.device, "bilinear_resize")
    lib().call("e4s_bilinear_resize", _p(out), _p(x), bs * C, h, w, size[0], size[1], 1 if align_corners else 0, _stream())
    return out


def maxpool3x3s2(x: torch.Tensor) -> torch.Tensor:
    x = _c(x, "input")
    bs, C, h, w = x.shape
    out = torch.empty((bs, C, (h - 1) // 2 + 1, (w - 1) // 2 + 1), dtype=torch.float32, device=x.device)
    lib().call("e4s_maxpool3x3s2", _p(out), _p(x), bs * C, h, w, _stream())
    return out


def gate_add_upsample(feat, gate=None, add_map=None, add_vec=None, up: int = 1) -> torch.Tensor:
    feat = _c(feat, "feat")
    bs, C, h, w = feat.shape
    out = torch.empty((bs, C, h * up, w * up), dtype=torch.float32, device=feat.device)
    am = _c(add_map, "add_map") if add_map is not None else None
    if am is not None and tuple(am.shape) != tuple(feat.shape):
        raise ValueError("add_map must have the shape of feat")
    lib().call("e4s_gate_add_upsample", _p(out), _p(feat), _p(gate), _p(am), _p(add_vec), bs * C, h, w, up, _stream())
    return out


def bilinear_argmax(logits: torch.Tensor, size, lut: Optional[torch.Tensor] = None) -> torch.Tensor:
    logits = _c(logits, "logits")
    bs, ncls, h, w = logits.shape
    out = torch.empty((bs, size[0], size[1]), dtype=torch.uint8, device=logits.device)
    lib().call("e4s_bilinear_argmax", _p(out), _p(logits), _p(lut), bs, ncls, h, w, size[0], size[1], _stream())
    return out


def bicubic_down_normalize(img01: torch.Tensor, taps: torch.Tensor, factor: int, mean: Optional[torch.Tensor] = None,
                           std: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None, pm1: bool = False) -> torch.Tensor:
    """``pm1``: the image is in [-1, 1] and ``(img + 1) / 2`` is applied on load (the same values; saves the pass that makes the [0, 1] copy)."""
    x = _c(img01, "image")
    bs, C, h, w = x.shape
    out = _out_like(out, (bs, C, h // factor, w // factor), x.device, "bicubic_down_normalize")
    lib().call("e4s_bicubic_down_normalize_pm1" if pm1 else "e4s_bicubic_down_normalize", _p(out), _p(x),
               _p(_c(taps, "taps")) if taps is not None else None, _p(mean), _p(std), bs, C, h, w, factor, _stream())
    return out


def tensor2im_u8(img: torch.Tensor) -> torch.Tensor:
    """``[bs, 3, H, W]`` float -> uint8 ``[bs, H, W, 3]`` with the reference's ``tensor2im`` arithmetic (truncating cast)."""
    x = _c(img, "image")
    bs, c, h, w = x.shape
    if c != 3:
        raise ValueError("tensor2im_u8 expects 3 channels")
    out = torch.empty((bs, h, w, 3), dtype=torch.uint8, device=x.device)
    lib().call("e4s_tensor2im_u8", _p(out), _p(x), bs, h, w, _stream())
    return out


__all__ = ['MxOperandMap', 'conv3x3_s1_c4_pair', 'ACT_NONE', 'ACT_RELU', 'ACT_SIGMOID', 'PreparedConv', '_is_f16x3', 'conv2d', 'PreparedWinograd', 'winograd_route', 'mx4_eligible', 'mx_conv_eligible', 'conv3x3_mx', 'conv3x3_s2_mx', 'conv3x3_s2_takes_mx', '_ShapeOnly', 'conv3x3_s2', 'conv3x3_s1_takes_mx3', 'conv3x3_s1', 'conv2d_winograd', 'plane_stats', 'vec_fc', 'se_gate', '_half_gates', 'half_gate', 'norm_gate_add', 'masked_avg_pool', '_out_like', 'bilinear_resize', 'maxpool3x3s2', 'gate_add_upsample', 'bilinear_argmax', 'bicubic_down_normalize', 'tensor2im_u8']
