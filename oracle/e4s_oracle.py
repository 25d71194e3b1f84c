"""CPU oracle for the E4S hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, in plain fp32 PyTorch-CPU functional code over a flat
``state_dict``, the arithmetic of the reference's regional-GAN-inversion path
(SURVEY §8a rows a1–a10).  It is the *checker*: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.
The product (``e4s2024_amd``) never imports it and has no CPU fallback.

Form: **faithful** — masked layers run the modulated convolution once per region and
sum ``out_i * segmap_i`` exactly as the reference does (``models/stylegan2/model.py:385-400,
442-456``), with separate blur / noise / bias-act steps, so its cost on the CPU is the
reference's cost and it doubles as the ``cpu_baseline`` ("port").

Pinning: the reference has no tests or golden vectors for this path, and its
dense arithmetic is PyTorch ATen (``torch==2.0.1`` pinned in ``requirements.txt:215``), which
is not under ``/root/reference``.  The oracle is therefore pinned against *outputs of the
reference itself run in the build container* (CPU, torch 2.10, through
``tests/golden/reference_shim.py``) — the fixtures in ``tests/golden/*.npz`` made by
``tests/golden/make_golden.py``; ``tests/test_oracle_golden.py`` checks every one of them.

Each function cites the reference lines it follows (paths relative to ``/root/reference``).
"""
from __future__ import annotations

import math
from typing import Dict, List, Mapping, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Mapping[str, Tensor]

SQRT2 = 2.0 ** 0.5


# =============================================================================== a1
def fused_leaky_relu(x: Tensor, bias: Optional[Tensor], negative_slope: float = 0.2, scale: float = SQRT2) -> Tensor:
    """``y = leaky_relu(x + b[c]) * scale`` with the bias broadcast along dim 1.
    models/stylegan2/op/fused_bias_act_kernel.cu:26-47 (act*10+grad == 30),
    models/stylegan2/op/fused_act.py:50-59, 84-85."""
    if bias is not None and bias.numel():
        x = x + bias.view((1, -1) + (1,) * (x.ndim - 2))
    return torch.where(x > 0, x, x * negative_slope) * scale


def fused_leaky_relu_backward(grad_out: Tensor, out: Tensor, negative_slope: float = 0.2, scale: float = SQRT2) -> Tuple[Tensor, Tensor]:
    """Backward of a1: ``grad_in = grad_out * (out > 0 ? 1 : slope) * scale`` using the
    *output* sign as reference, ``grad_bias`` = sum over all dims but 1.
    fused_bias_act_kernel.cu:43 (case 31), fused_act.py:18-38."""
    gi = torch.where(out > 0, grad_out, grad_out * negative_slope) * scale
    dims = [0] + list(range(2, gi.ndim))
    return gi, gi.sum(dims)


# =============================================================================== a2
def upfirdn2d(x: Tensor, kernel: Tensor, up: int = 1, down: int = 1, pad: Tuple[int, int] = (0, 0)) -> Tensor:
    """Upsample (zero insertion) → pad (negative pad crops) → true 2-D convolution with
    ``kernel`` (i.e. correlation with the flipped kernel) → decimate.  NCHW in/out.
    models/stylegan2/op/upfirdn2d.py:85-147 (shape rule :100-101),
    upfirdn2d_kernel.cu:71-81 (flipped taps) and :100-131; CPU form of the same op in
    swap_face_fine/gpen/face_model/op/upfirdn2d.py:160-194."""
    return upfirdn2d_xy(x, kernel, up, up, down, down, pad[0], pad[1], pad[0], pad[1])


def upfirdn2d_xy(x: Tensor, kernel: Tensor, up_x: int, up_y: int, down_x: int, down_y: int,
                 pad_x0: int, pad_x1: int, pad_y0: int, pad_y1: int) -> Tensor:
    n, c, h, w = x.shape
    kh, kw = kernel.shape
    z = x.new_zeros(n, c, h * up_y, w * up_x)
    z[:, :, ::up_y, ::up_x] = x                                   # value at index i*up
    z = F.pad(z, [max(pad_x0, 0), max(pad_x1, 0), max(pad_y0, 0), max(pad_y1, 0)])
    z = z[:, :, max(-pad_y0, 0): z.shape[2] - max(-pad_y1, 0), max(-pad_x0, 0): z.shape[3] - max(-pad_x1, 0)]
    wgt = torch.flip(kernel, [0, 1]).reshape(1, 1, kh, kw).to(x.dtype)
    zz = z.reshape(n * c, 1, z.shape[2], z.shape[3])
    o = F.conv2d(zz, wgt)
    o = o.reshape(n, c, o.shape[2], o.shape[3])
    return o[:, :, ::down_y, ::down_x].contiguous()


def make_blur_kernel(k1d: Sequence[float] = (1, 3, 3, 1), gain: float = 1.0) -> Tensor:
    """models/stylegan2/model.py:23-31 (``make_kernel``) times ``gain``
    (``factor**2`` in Upsample :39 and Blur(upsample_factor=2) :84-85)."""
    k = torch.tensor(list(k1d), dtype=torch.float32)
    k2 = k[None, :] * k[:, None]
    return k2 / k2.sum() * gain


# =============================================================================== a3
def equal_linear(x: Tensor, weight: Tensor, bias: Optional[Tensor], lr_mul: float = 1.0, activation: bool = False) -> Tensor:
    """models/stylegan2/model.py:135-164."""
    scale = (1.0 / math.sqrt(weight.shape[1])) * lr_mul
    if activation:
        return fused_leaky_relu(F.linear(x, weight * scale), bias * lr_mul)
    return F.linear(x, weight * scale, None if bias is None else bias * lr_mul)


def modulated_conv2d(x: Tensor, style: Tensor, weight: Tensor, mod_weight: Tensor, mod_bias: Tensor,
                     demodulate: bool = True, upsample: bool = False, blur_kernel: Optional[Tensor] = None) -> Tensor:
    """Fused branch of ``ModulatedConv2d.forward`` — models/stylegan2/model.py:276-320.
    ``weight`` is the parameter ``[1, Cout, Cin, k, k]``; ``style`` is ``[bs, 512]``."""
    bs, cin, h, w = x.shape
    _, cout, _, k, _ = weight.shape
    scale = 1.0 / math.sqrt(cin * k * k)                                        # :223-224
    s = equal_linear(style, mod_weight, mod_bias).view(bs, 1, cin, 1, 1)       # :276
    wt = scale * weight * s                                                     # :277
    if demodulate:
        d = torch.rsqrt(wt.pow(2).sum([2, 3, 4]) + 1e-8)                        # :280
        wt = wt * d.view(bs, cout, 1, 1, 1)
    if upsample:
        xin = x.reshape(1, bs * cin, h, w)
        wtt = wt.transpose(1, 2).reshape(bs * cin, cout, k, k)                  # :289-294
        out = F.conv_transpose2d(xin, wtt, padding=0, stride=2, groups=bs)      # :295-297
        out = out.view(bs, cout, out.shape[2], out.shape[3])
        # Blur(pad=(pad0,pad1)) with factor 2 and k=3: p = (4-2)-(3-1) = 0 -> pad (1,1)   :206-213
        p = (blur_kernel.shape[0] - 2) - (k - 1)
        pad0, pad1 = (p + 1) // 2 + 2 - 1, p // 2 + 1
        return upfirdn2d(out, blur_kernel, pad=(pad0, pad1))                    # :300
    xin = x.reshape(1, bs * cin, h, w)
    out = F.conv2d(xin, wt.view(bs * cout, cin, k, k), padding=k // 2, groups=bs)  # :313-316
    return out.view(bs, cout, out.shape[2], out.shape[3])


# =========================================================================== a4 / a5
def nearest_mask(mask: Tensor, size: Tuple[int, int]) -> Tensor:
    """``F.interpolate(mask, size, mode='nearest')`` — models/stylegan2/model.py:391, 447."""
    return F.interpolate(mask, size=size, mode="nearest")


def styled_conv(sd: SD, p: str, x: Tensor, style: Tensor, mask: Optional[Tensor], noise: Tensor,
                masked: bool, upsample: bool) -> Tensor:
    """``StyledConv.forward`` — models/stylegan2/model.py:382-423.  ``p`` is the key prefix
    (e.g. ``"G.convs.0."``); ``style`` is ``[bs,12,512]`` when ``masked`` else ``[bs,512]``;
    ``noise`` is the explicit ``[1 or bs,1,H,W]`` map (the oracle never draws noise)."""
    W, mw, mb = sd[p + "conv.weight"], sd[p + "conv.modulation.weight"], sd[p + "conv.modulation.bias"]
    bk = sd.get(p + "conv.blur.kernel") if upsample else None
    if not masked:
        out = modulated_conv2d(x, style, W, mw, mb, True, upsample, bk)
    else:
        bs, _, h, w = x.shape
        ho, wo = (h * 2, w * 2) if upsample else (h, w)                                  # :389
        seg = nearest_mask(mask, (ho, wo))                                               # :391
        out = x.new_zeros(bs, W.shape[1], ho, wo)
        for c in range(style.shape[1]):                                                  # :395-398
            out = out + modulated_conv2d(x, style[:, c], W, mw, mb, True, upsample, bk) * seg[:, c:c + 1]
    out = out + sd[p + "noise.weight"] * noise                                           # :335, 419
    return fused_leaky_relu(out, sd[p + "activate.bias"])                                # :421


def to_rgb(sd: SD, p: str, x: Tensor, style: Tensor, mask: Optional[Tensor], skip: Optional[Tensor], masked: bool) -> Tensor:
    """``ToRGB.forward`` — models/stylegan2/model.py:439-479 (1×1, ``demodulate=False`` :434)."""
    W, mw, mb = sd[p + "conv.weight"], sd[p + "conv.modulation.weight"], sd[p + "conv.modulation.bias"]
    if not masked:
        out = modulated_conv2d(x, style, W, mw, mb, False, False, None)
    else:
        bs, _, h, w = x.shape
        seg = nearest_mask(mask, (h, w))                                                 # :447
        out = x.new_zeros(bs, 3, h, w)
        for c in range(style.shape[1]):                                                  # :451-454
            out = out + modulated_conv2d(x, style[:, c], W, mw, mb, False, False, None) * seg[:, c:c + 1]
    out = out + sd[p + "bias"]                                                           # :472
    if skip is not None:
        out = out + upfirdn2d(skip, sd[p + "upsample.kernel"], up=2, down=1, pad=(2, 1))  # :42-53, 475
    return out


# =============================================================================== a6
def generator_masked_flags(size: int, remaining_layer_idx: int):
    """mask_op flags of Generator.__init__ — models/stylegan2/model.py:549-579."""
    log_size = int(math.log2(size))
    conv_masked, rgb_masked = [], []
    for i in range(3, log_size + 1):
        m = not (i > (2 + remaining_layer_idx // 2))                                     # :560, 568
        conv_masked += [m, m]
        rgb_masked.append(not (remaining_layer_idx != 17 and i >= (2 + remaining_layer_idx // 2)))  # :576
    return conv_masked, rgb_masked


def generator_forward(sd: SD, codes: Tensor, mask: Tensor, noise: Optional[List[Tensor]] = None, *, size: int = 1024,
                      remaining_layer_idx: int = 13, split_layer_idx: int = 5, prefix: str = "G.") -> Tuple[Tensor, Tensor]:
    """``Generator.forward(styles=[codes], ..., input_is_latent=True, use_structure_code=False)``
    with ``codes`` of shape ``[bs, n_cls, n_latent, 512]`` — models/stylegan2/model.py:607-698.
    ``noise=None`` means the registered buffers (``randomize_noise=False``, :628-630).
    Returns ``(image, intermediate_feats)``."""
    log_size = int(math.log2(size))
    num_layers = (log_size - 2) * 2 + 1
    if noise is None:
        noise = [sd[f"{prefix}noises.noise_{i}"] for i in range(num_layers)]
    conv_masked, rgb_masked = generator_masked_flags(size, remaining_layer_idx)
    latent = codes                                                                       # :645-649
    bs = latent.shape[0]
    out = sd[prefix + "input.input"].repeat(bs, 1, 1, 1)                                 # :346, 661
    out = styled_conv(sd, prefix + "conv1.", out, latent[:, :, 0], mask, noise[0], True, False)     # :662
    skip = to_rgb(sd, prefix + "to_rgb1.", out, latent[:, :, 1], mask, None, True)       # :663
    feats = None
    i = 1
    for j in range(log_size - 2):                                                        # :666-690
        c1, c2, rgb = f"{prefix}convs.{2 * j}.", f"{prefix}convs.{2 * j + 1}.", f"{prefix}to_rgbs.{j}."
        n1, n2 = noise[1 + 2 * j], noise[2 + 2 * j]
        if i < remaining_layer_idx:
            out = styled_conv(sd, c1, out, latent[:, :, i] if conv_masked[2 * j] else latent[:, 0, i], mask, n1, conv_masked[2 * j], True)
            if i + 2 == split_layer_idx:
                feats = out                                                              # :673-678
            out = styled_conv(sd, c2, out, latent[:, :, i + 1] if conv_masked[2 * j + 1] else latent[:, 0, i + 1], mask, n2, conv_masked[2 * j + 1], False)
            if remaining_layer_idx == 17 or i + 2 != remaining_layer_idx:                # :681-684
                skip = to_rgb(sd, rgb, out, latent[:, :, i + 2] if rgb_masked[j] else latent[:, 0, i + 2], mask, skip, rgb_masked[j])
            else:
                skip = to_rgb(sd, rgb, out, latent[:, 0, i + 2], mask, skip, rgb_masked[j])
        else:                                                                            # :686-688
            out = styled_conv(sd, c1, out, latent[:, 0, i], mask, n1, conv_masked[2 * j], True)
            out = styled_conv(sd, c2, out, latent[:, 0, i + 1], mask, n2, conv_masked[2 * j + 1], False)
            skip = to_rgb(sd, rgb, out, latent[:, 0, i + 2], mask, skip, rgb_masked[j])
        i += 2
    return skip, feats


# =============================================================================== a7
def cal_style_codes(sd: SD, style_vectors: Tensor, latent_avg: Tensor, remaining_layer_idx: int = 13) -> Tensor:
    """``Net3.cal_style_codes`` with ``start_from_latent_avg=True, learn_in_w=False`` —
    models/networks.py:223-253; ``LocalMLP`` :23-49 (``nn.LeakyReLU()`` slope 0.01 :34)."""
    bs, ncls, _ = style_vectors.shape
    nw = remaining_layer_idx if remaining_layer_idx != 17 else 18
    codes = []
    for c in range(ncls):
        p = f"MLPs.{c}.mlp."
        h = equal_linear(style_vectors[:, c], sd[p + "0.weight"], sd[p + "0.bias"])
        h = F.leaky_relu(h, 0.01)
        h = equal_linear(h, sd[p + "2.weight"], sd[p + "2.bias"])
        codes.append(h.view(bs, nw, 512))
    codes = torch.stack(codes, dim=1)                                                    # [bs,ncls,nw,512]
    if remaining_layer_idx != 17:
        codes = codes + latent_avg[:remaining_layer_idx][None, None]                     # :247
        rest = latent_avg[remaining_layer_idx:][None, None].expand(bs, ncls, -1, -1)     # :248
        return torch.cat([codes, rest], dim=2)
    return codes + latent_avg[None, None]


# =============================================================================== a8
def instance_norm(x: Tensor, eps: float = 1e-5) -> Tensor:
    """``InstanceNorm2d(C)`` defaults: no affine, no running stats, biased variance."""
    m = x.mean(dim=(2, 3), keepdim=True)
    v = ((x - m) ** 2).mean(dim=(2, 3), keepdim=True)
    return (x - m) / torch.sqrt(v + eps)


ENCODER_UNITS = ([(64, 128, 2)] + [(128, 128, 1)] * 2 + [(128, 256, 2)] + [(256, 256, 1)] * 3 +
                 [(256, 512, 2)] + [(512, 512, 1)] * 13 + [(512, 512, 2)] + [(512, 512, 1)] * 2)
"""(in, depth, stride) of the 24 units — models/encoders/psp_encoders.py:323-328, helpers.py:23-24."""


def encoder_unit(sd: SD, p: str, x: Tensor, cin: int, depth: int, stride: int) -> Tensor:
    """``bottleneck_IR_SE_Ours`` — models/encoders/helpers.py:122-144, ``SEModule`` :56-72."""
    if cin == depth:
        sc = x[:, :, ::stride, ::stride]                                                 # MaxPool2d(1, stride)
    else:
        sc = instance_norm(F.conv2d(x, sd[p + "shortcut_layer.0.weight"], stride=stride))
    r = instance_norm(x)
    r = F.conv2d(r, sd[p + "res_layer.1.weight"], padding=1)
    r = F.prelu(r, sd[p + "res_layer.2.weight"])
    r = F.conv2d(r, sd[p + "res_layer.3.weight"], stride=stride, padding=1)
    r = instance_norm(r)
    g = r.mean(dim=(2, 3), keepdim=True)
    g = F.relu(F.conv2d(g, sd[p + "res_layer.5.fc1.weight"]))
    g = torch.sigmoid(F.conv2d(g, sd[p + "res_layer.5.fc2.weight"]))
    return r * g + sc


def masked_avg_pool(feats: Tensor, mask: Tensor) -> Tensor:
    """``get_per_comp_styleCode`` — models/encoders/psp_encoders.py:355-375."""
    seg = F.interpolate(mask, size=feats.shape[2:], mode="nearest")
    bs, f = feats.shape[:2]
    out = feats.new_zeros(bs, seg.shape[1], f)
    for b in range(bs):
        for c in range(seg.shape[1]):
            m = seg[b, c] != 0
            area = int(m.sum())
            if area > 0:
                out[b, c] = feats[b][:, m].mean(1)
    return out


def encoder_forward(sd: SD, x256: Tensor, mask: Tensor, prefix: str = "encoder.") -> Tuple[Tensor, Tensor]:
    """``FSEncoder_PSP.forward`` — models/encoders/psp_encoders.py:377-401."""
    x = F.conv2d(x256, sd[prefix + "input_layer.0.weight"], padding=1)                   # :334-336
    x = F.prelu(instance_norm(x), sd[prefix + "input_layer.2.weight"])
    taps = {}
    for i, (cin, depth, stride) in enumerate(ENCODER_UNITS):
        x = encoder_unit(sd, f"{prefix}body.{i}.", x, cin, depth, stride)
        if i in (6, 20, 23):
            taps[i] = x
    vec = torch.cat([masked_avg_pool(taps[6], mask), masked_avg_pool(taps[20], mask), masked_avg_pool(taps[23], mask)], dim=2)
    return vec, torch.zeros_like(x)                                                      # :392


def get_style_vectors(sd: SD, img: Tensor, mask: Tensor) -> Tuple[Tensor, Tensor]:
    """``Net3.get_style_vectors`` — models/networks.py:206-221 (bilinear, align_corners=False)."""
    return encoder_forward(sd, F.interpolate(img, (256, 256), mode="bilinear"), mask)


# =============================================================================== a9
def _bn(sd: SD, p: str, x: Tensor) -> Tensor:
    return F.batch_norm(x, sd[p + "running_mean"], sd[p + "running_var"], sd[p + "weight"], sd[p + "bias"], False, 0.0, 1e-5)


def _conv_bn_relu(sd: SD, p: str, x: Tensor, stride: int = 1, padding: int = 1) -> Tensor:
    """``ConvBNReLU`` — swap_face_fine/face_parsing/model.py:20-35."""
    return F.relu(_bn(sd, p + "bn.", F.conv2d(x, sd[p + "conv.weight"], stride=stride, padding=padding)))


def _basic_block(sd: SD, p: str, x: Tensor, stride: int) -> Tensor:
    """``BasicBlock`` — swap_face_fine/face_parsing/resnet.py:21-49."""
    r = F.relu(_bn(sd, p + "bn1.", F.conv2d(x, sd[p + "conv1.weight"], stride=stride, padding=1)))
    r = _bn(sd, p + "bn2.", F.conv2d(r, sd[p + "conv2.weight"], padding=1))
    sc = x
    if (p + "downsample.0.weight") in sd:
        sc = _bn(sd, p + "downsample.1.", F.conv2d(x, sd[p + "downsample.0.weight"], stride=stride))
    return F.relu(sc + r)


def resnet18_forward(sd: SD, x: Tensor, p: str = "cp.resnet.") -> Tuple[Tensor, Tensor, Tensor]:
    """``Resnet18.forward`` — resnet.py:72-81."""
    x = F.relu(_bn(sd, p + "bn1.", F.conv2d(x, sd[p + "conv1.weight"], stride=2, padding=3)))
    x = F.max_pool2d(x, 3, 2, 1)
    for b in range(2):
        x = _basic_block(sd, f"{p}layer1.{b}.", x, 1)
    feats = []
    for li in (2, 3, 4):
        for b in range(2):
            x = _basic_block(sd, f"{p}layer{li}.{b}.", x, 2 if b == 0 else 1)
        feats.append(x)
    return tuple(feats)


def _arm(sd: SD, p: str, x: Tensor) -> Tensor:
    """``AttentionRefinementModule`` — face_parsing/model.py:73-89."""
    feat = _conv_bn_relu(sd, p + "conv.", x)
    a = feat.mean(dim=(2, 3), keepdim=True)
    a = torch.sigmoid(_bn(sd, p + "bn_atten.", F.conv2d(a, sd[p + "conv_atten.weight"])))
    return feat * a


def bisenet_forward(sd: SD, x: Tensor, aux: bool = False):
    """``BiSeNet.forward`` — face_parsing/model.py:247-260 (ContextPath :110-131, FFM :206-216,
    BiSeNetOutput :49-53).  Returns the main head's logits at input size (and the two aux
    heads when ``aux``)."""
    H, W = x.shape[2:]
    f8, f16, f32 = resnet18_forward(sd, x)
    avg = _conv_bn_relu(sd, "cp.conv_avg.", f32.mean(dim=(2, 3), keepdim=True), padding=0)
    f32s = _arm(sd, "cp.arm32.", f32) + avg                                              # nearest up of 1x1 = broadcast
    f32u = _conv_bn_relu(sd, "cp.conv_head32.", F.interpolate(f32s, f16.shape[2:], mode="nearest"))
    f16s = _arm(sd, "cp.arm16.", f16) + f32u
    f16u = _conv_bn_relu(sd, "cp.conv_head16.", F.interpolate(f16s, f8.shape[2:], mode="nearest"))
    fcat = torch.cat([f8, f16u], dim=1)
    feat = _conv_bn_relu(sd, "ffm.convblk.", fcat, padding=0)
    a = feat.mean(dim=(2, 3), keepdim=True)
    a = torch.sigmoid(F.conv2d(F.relu(F.conv2d(a, sd["ffm.conv1.weight"])), sd["ffm.conv2.weight"]))
    fuse = feat * a + feat

    def head(p, t):
        return F.conv2d(_conv_bn_relu(sd, p + "conv.", t), sd[p + "conv_out.weight"])

    out = F.interpolate(head("conv_out.", fuse), (H, W), mode="bilinear", align_corners=True)
    if not aux:
        return out
    o16 = F.interpolate(head("conv_out16.", f16u), (H, W), mode="bilinear", align_corners=True)
    o32 = F.interpolate(head("conv_out32.", f32u), (H, W), mode="bilinear", align_corners=True)
    return out, o16, o32


# ============================================================================== a10
SEG_MEAN = (0.485, 0.456, 0.406)
SEG_STD = (0.229, 0.224, 0.225)
"""face_parsing/model.py:15-16."""


def bicubic_taps(factor: int, a: float = -0.5) -> Tensor:
    """1-D taps of ``BicubicDownSample`` — face_parsing_demo.py:16-36."""
    size = factor * 4
    xs = (torch.arange(size, dtype=torch.float32) - math.floor(size / 2) + 0.5) / factor
    ax = xs.abs()
    k = torch.where(ax <= 1.0, (a + 2.0) * ax ** 3 - (a + 3.0) * ax ** 2 + 1.0,
                    torch.where(ax < 2.0, a * ax ** 3 - 5.0 * a * ax ** 2 + 8.0 * a * ax - 4.0 * a, torch.zeros_like(ax)))
    return k / k.sum()


def bicubic_downsample(x: Tensor, factor: int = 2) -> Tensor:
    """``BicubicDownSample.forward`` (reflect padding, vertical then horizontal pass) —
    face_parsing_demo.py:46-84."""
    k = bicubic_taps(factor)
    size = factor * 4
    padt = (size - factor) // 2
    padb = (size - factor) - padt
    c = x.shape[1]
    x = F.pad(x, (0, 0, padt, padb), "reflect")
    x = F.conv2d(x, k.view(1, 1, size, 1).repeat(c, 1, 1, 1), stride=(factor, 1), groups=c)
    x = F.pad(x, (padt, padb, 0, 0), "reflect")
    x = F.conv2d(x, k.view(1, 1, 1, size).repeat(c, 1, 1, 1), stride=(1, factor), groups=c)
    return x


def parser_preprocess(img01: Tensor) -> Tensor:
    """``FaceParser.preprocess_img`` on a ``[bs,3,1024,1024]`` tensor in [0,1] (the >=512 branch):
    bicubic ↓2, clamp, ImageNet normalise — face_parsing_demo.py:151-156."""
    x = bicubic_downsample(img01, img01.shape[-1] // 512).clamp(0, 1)
    m = torch.tensor(SEG_MEAN).view(1, 3, 1, 1)
    s = torch.tensor(SEG_STD).view(1, 3, 1, 1)
    return (x - m) / s


def parse_labels(sd: SD, x: Tensor) -> Tensor:
    """``FaceParser.forward`` after preprocessing: argmax over the 19 logits, first index wins
    ties — face_parsing_demo.py:168-170.  Returns uint8 ``[bs,H,W]``."""
    return torch.argmax(bisenet_forward(sd, x), dim=1).to(torch.uint8)


_REMAP_19_TO_12 = np.zeros(256, dtype=np.uint8)
for _src, _dst in {0: 0, 12: 1, 13: 1, 2: 2, 3: 2, 4: 3, 5: 3, 17: 4, 10: 5, 1: 6, 7: 7, 8: 7, 14: 8, 11: 9, 6: 10, 9: 11}.items():
    _REMAP_19_TO_12[_src] = _dst


def remap_19_to_12(labels: np.ndarray) -> np.ndarray:
    """``__ffhq_masks_to_faceParser_mask_detailed`` — datasets/dataset.py:58-108 (every label not
    listed, i.e. 15, 16, 18, stays 0)."""
    return _REMAP_19_TO_12[labels]


# ================================================================== boundary helpers
def label_map_to_onehot(label: Tensor, num_cls: int) -> Tensor:
    """``labelMap2OneHot`` — utils/torch_utils.py:207-213.  ``label`` int64 ``[bs,1,H,W]``."""
    bs, _, h, w = label.shape
    return torch.zeros(bs, num_cls, h, w).scatter_(1, label, 1.0)


def tensor2im_array(img: Tensor) -> np.ndarray:
    """``tensor2im`` up to the PIL wrap — utils/torch_utils.py:64-76: ``(x+1)/2`` clamp ×255,
    **truncating** cast.  ``img`` is ``[3,H,W]``; returns uint8 ``[H,W,3]``."""
    v = img.permute(1, 2, 0).detach().cpu().numpy()
    v = (v + 1) / 2
    v[v < 0] = 0
    v[v > 1] = 1
    return (v * 255).astype("uint8")


# ======================================================================== work counts
def synthesis_flops(size: int = 1024) -> float:
    """Algorithmic FLOPs of one ``gen_img`` sample: every modulated conv counted once
    (148.52 GFLOP at 1024)."""
    ch = {4: 512, 8: 512, 16: 512, 32: 512, 64: 512, 128: 256, 256: 128, 512: 64, 1024: 32}
    fl = 2 * 512 * 512 * 9 * 16 + 2 * 512 * 3 * 16
    cin = 512
    r = 8
    while r <= size:
        co = ch[r]
        fl += 2 * cin * co * 9 * (r // 2) ** 2      # stride-2 transposed conv, per input pixel
        fl += 2 * co * co * 9 * r * r
        fl += 2 * co * 3 * r * r
        cin = co
        r *= 2
    return float(fl)


# --------------------------------------------------------------------------- f2 / f3: mask surgery between parse and synthesis
_BG_CLASSES = (4, 0, 8, 7, 11)      # hair, background, neck, ear, ear-ring  (swap_face_mask.py:210-220)


def swap_head_mask_hole_first(source: np.ndarray, target: np.ndarray):
    """``swap_head_mask_hole_first`` — swap_face_fine/swap_face_mask.py:194-333, vectorised but assignment for assignment
    (later writes win).  ``source`` / ``target``: integer 12-class maps ``[H, W]``.  Returns
    ``(res, hole_mask, hole_map, nose_line)`` like the reference, plus ``eye_line`` as a fifth value."""
    source = np.asarray(source)
    target = np.asarray(target)
    H, W = target.shape
    src_face = ~np.isin(source, _BG_CLASSES)                                             # :210-214
    tgt_face = ~np.isin(target, _BG_CLASSES)                                             # :216-220
    hole = np.logical_xor(np.logical_and(src_face, tgt_face), tgt_face)                  # :222-223
    eye_line, nose_line = int(2 / 5 * H), int(3 / 5 * H)                                 # :232-233
    if np.any(source == 3):
        eye_line = int(np.where(source == 3)[0].max())                                   # :234-235
    elif np.any(source == 2):
        eye_line = int(np.where(source == 2)[0].max())                                   # :236-237
    if np.any(source == 5):
        nose_line = int(np.where(source == 5)[0].max())                                  # :238-239
    if len(hole) >= eye_line:
        hole[:eye_line, :] = False                                                       # :243-244
    res = np.zeros_like(target)
    res[target == 0] = 99                                                                # :247-251
    res[target == 8] = 8
    res[target == 7] = 7
    res[target == 11] = 11
    res[source == 1] = 1                                                                 # :263-269
    res[source == 2] = 2
    res[np.logical_and(source == 4, target == 2)] = 2
    res[source == 3] = 3
    res[source == 5] = 5
    res[source == 6] = 6
    res[source == 9] = 9
    rows = np.arange(H)[:, None]
    skin_rows = np.where(target == 6, rows, 0).astype(np.int64)                          # :282-285: skin in row 0 counts as "none"
    skin_rows[skin_rows == 0] = H
    skin_top = skin_rows.min(axis=0)                                                     # :286
    fg = np.logical_and(target == 0, np.logical_and(rows <= skin_top[None, :], skin_top[None, :] != H))   # :287-299
    res[fg] = 98                                                                         # :300-301
    res[target == 4] = 4                                                                 # :304-305
    res[target == 10] = 10
    res[res == 0] = 6                                                                    # :310-312
    res[res == 99] = 0
    res[res == 98] = 0
    hole_map = res.copy()
    hole_map[hole] = 17                                                                  # :313-314
    return res, hole, hole_map, nose_line, eye_line


def swap_comp_style_vector(style_vectors1: Tensor, style_vectors2: Tensor, comp_indices: Sequence[int] = (),
                           below_face_interpolation: bool = False) -> Tensor:
    """``swap_comp_style_vector`` — swap_face_fine/swap_face_mask.py:336-367.  ``style_vectors1`` = the TARGET frame's per-region style
    vectors ``[1, 12, D]``, ``style_vectors2`` = the driven (source) face's.  Assignment for assignment:

        :346-349  the listed components come from the source
        :355      ears (7) = mean of both, unconditionally (the ``if`` above it is commented out in the reference)
        :358      ear-rings (11) always the target's
        :361-362  neck (8) = mean of both when ``belowFace_interpolation``
        :365-366  teeth (9) the target's when ``torch.sum(source[:, 9, :]) == 0`` (an empty region's vector is exactly zero:
                  models/encoders/psp_encoders.py:368-370)

    The reference is only ever called with batch 1 (face_swap_video_pipeline.py:429-436); for a batch the teeth rule (a sum over the whole
    ``[:, 9, :]`` slice) is applied here per sample, i.e. a batch behaves as that many batch-1 calls."""
    if style_vectors1.shape != style_vectors2.shape or style_vectors1.dim() != 3:
        raise ValueError("style vectors must both be [bs, n_comp, D]")
    out = style_vectors1.clone()
    for c in comp_indices:
        out[:, c, :] = style_vectors2[:, c, :]
    out[:, 7, :] = (style_vectors1[:, 7, :] + style_vectors2[:, 7, :]) / 2
    out[:, 11, :] = style_vectors1[:, 11, :]
    if below_face_interpolation:
        out[:, 8, :] = (style_vectors1[:, 8, :] + style_vectors2[:, 8, :]) / 2
    for b in range(out.shape[0]):
        if torch.sum(style_vectors2[b: b + 1, 9, :]) == 0:
            out[b, 9, :] = style_vectors1[b, 9, :]
    return out


def erode_mask(mask: np.ndarray, radius: int = 3) -> np.ndarray:
    """``erode_mask(mask, img, radius)[0]`` — training/video_swap_ft_coach.py:64-93.  ``mask``: integer 12-class map ``[H, W]``.
    face = not {0, 4, 11} (:72-73); ``cv2.erode`` with a ``(2r+1)^2`` box of ones, ``BORDER_CONSTANT`` 0 (:79-81) = a flat erosion whose
    outside-the-image pixels count as not-face; the label survives where the eroded face mask holds (:83-84).
    cv2 is neither in the reference tree nor in this image: **pinned by definition only** (a flat erosion is a logical AND over the window;
    checked against scipy.ndimage.binary_erosion, an independent implementation, in tests/test_oracle_golden.py)."""
    face = ~np.isin(mask, (0, 4, 11))
    h, w = face.shape
    pad = np.zeros((h + 2 * radius, w + 2 * radius), dtype=bool)
    pad[radius: radius + h, radius: radius + w] = face
    er = np.ones((h, w), dtype=bool)
    for dy in range(2 * radius + 1):
        for dx in range(2 * radius + 1):
            er &= pad[dy: dy + h, dx: dx + w]
    out = np.zeros_like(mask)
    out[er] = mask[er]
    return out


def frames_to_tensor(frames_u8: np.ndarray) -> Tensor:
    """``Compose([ToTensor(), Normalize((.5,.5,.5), (.5,.5,.5))])`` — datasets/dataset.py:32, 45 (face_swap_video_pipeline.py:338-339) on uint8
    ``[bs, H, W, 3]``: torchvision's ToTensor is ``permute -> float32 -> div(255)``, Normalize ``(x - mean) / std``."""
    t = torch.from_numpy(np.ascontiguousarray(frames_u8)).permute(0, 3, 1, 2).to(torch.float32).div(255)
    return t.sub(0.5).div(0.5)


def _flat_morph(mask: np.ndarray, radius: int, op) -> np.ndarray:
    """Flat (2r+1)^2 dilation / erosion with the 'geodesic' border of utils/morphology.py:76-81, 150-155: pixels outside the image
    are ignored.  ``mask``: float ``[..., H, W]``."""
    H, W = mask.shape[-2:]
    pad_val = -np.inf if op is np.maximum else np.inf
    p = np.pad(mask, [(0, 0)] * (mask.ndim - 2) + [(radius, radius), (radius, radius)], constant_values=pad_val)
    out = np.full_like(mask, pad_val)
    for dy in range(2 * radius + 1):
        for dx in range(2 * radius + 1):
            out = op(out, p[..., dy:dy + H, dx:dx + W])
    return out


def create_masks_expansion(mask: np.ndarray, radius: int):
    """``create_masks(mask, operation='expansion', radius)`` — gradio_utils/face_swapping.py:203-221:
    ``(content, border, full) = (mask, clip(dilate(mask) - erode(mask), 0, 1), dilate(mask))``."""
    mask = np.asarray(mask, dtype=np.float32)
    full = _flat_morph(mask, radius, np.maximum)
    ero = _flat_morph(mask, radius, np.minimum)
    return mask, np.clip(full - ero, 0, 1), full


def foreground_mask(swapped: np.ndarray, hole: np.ndarray) -> np.ndarray:
    """face_swap_video_pipeline.py:456-461: everything except background / ear-ring / ear / hair / neck, plus the hole."""
    fg = ~np.isin(swapped, (0, 11, 7, 4, 8))
    return np.logical_or(fg, hole).astype(np.float32)


# ------------------------------------------------------------------------------------ f3: multi-band (Laplacian pyramid) blend
# swap_face_fine/multi_band_blending.py:5-74, called per frame at face_swap_video_pipeline.py:473.  PARITY UNPINNED: the arithmetic is
# OpenCV's (cv2.pyrDown / cv2.pyrUp / cv2.add; opencv-python 4.7.0.72 in requirements.txt), which is neither under /root/reference nor
# installed in this image, and the reference holds no fixture for it.  What follows restates OpenCV's published algorithm
# (modules/imgproc/src/pyramids.cpp: 5x5 kernel [1 4 6 4 1]^2 / 256, BORDER_REFLECT_101, 8-bit results rounded as (sum + 128) >> 8;
# pyrUp = zero-insertion convolved with 4x that kernel, evaluated as [1 6 1]/8 at even and [4 4]/8 at odd destinations, with the last
# source pixel treated as (s[n-2] + 7 s[n-1]) / 8 and s[n-1]) and anchors on the reference's own call site for types and level bookkeeping.
_PYR_K = np.array([1.0, 4.0, 6.0, 4.0, 1.0])


def _reflect101(idx: np.ndarray, n: int) -> np.ndarray:
    if n == 1:
        return np.zeros_like(idx)
    idx = np.abs(idx)
    return np.where(idx >= n, 2 * (n - 1) - idx, idx)


def _pyr_down_axis(a: np.ndarray, axis: int) -> np.ndarray:
    n = a.shape[axis]
    out_n = (n + 1) // 2
    centers = 2 * np.arange(out_n)
    acc = 0
    for t, k in zip(range(-2, 3), _PYR_K):
        acc = acc + k * np.take(a, _reflect101(centers + t, n), axis=axis)
    return acc


def pyr_down(img: np.ndarray) -> np.ndarray:
    """cv2.pyrDown of an [H, W, C] image.  uint8 in -> uint8 out, rounded (sum + 128) >> 8; float in -> same float type, sum / 256."""
    a = img.astype(np.float64)
    s = _pyr_down_axis(_pyr_down_axis(a, 1), 0)
    if img.dtype == np.uint8:
        return np.clip(np.floor((s + 128.0) / 256.0), 0, 255).astype(np.uint8)
    return (s / 256.0).astype(img.dtype)


def _pyr_up_axis(a: np.ndarray, axis: int) -> np.ndarray:
    n = a.shape[axis]
    a = np.moveaxis(a, axis, 0)
    out = np.empty((2 * n,) + a.shape[1:], dtype=np.float64)
    if n == 1:
        out[0] = out[1] = 8.0 * a[0]
    else:
        prev = np.concatenate([a[1:2], a[:-1]])            # s[i-1], reflected at the first pixel
        nxt = np.concatenate([a[1:], a[-1:]])              # s[i+1]; the last pixel has its own rule below
        out[0::2] = prev + 6.0 * a + nxt
        out[1::2] = 4.0 * (a + nxt)
        out[2 * n - 2] = a[n - 2] + 7.0 * a[n - 1]
        out[2 * n - 1] = 8.0 * a[n - 1]
    return np.moveaxis(out, 0, axis)


def pyr_up(img: np.ndarray) -> np.ndarray:
    """cv2.pyrUp of an [H, W, C] float image to [2H, 2W, C]."""
    s = _pyr_up_axis(_pyr_up_axis(img.astype(np.float64), 1), 0) / 64.0
    return s.astype(img.dtype if img.dtype != np.uint8 else np.float64)


def laplacian_blend(a_u8: np.ndarray, b: np.ndarray, m: np.ndarray, num_levels: int = 10) -> np.ndarray:
    """``Laplacian_Pyramid_Blending_with_mask(A, B, m, num_levels)`` (multi_band_blending.py:5-48) with the types of its one caller
    (``blending`` :51-74, at 1024 x 1024): A uint8 (its Gaussian pyramid is rounded to 8 bits at every level), B float64, m float32."""
    ga, gb, gm = a_u8.copy(), b.copy(), m.copy()
    gpa, gpb, gpm = [ga], [gb], [gm]
    for _ in range(num_levels):
        ga, gb, gm = pyr_down(ga), pyr_down(gb), pyr_down(gm)
        gpa.append(ga.astype(np.float32)); gpb.append(gb.astype(np.float32)); gpm.append(gm.astype(np.float32))
    lpa, lpb, gmr = [gpa[num_levels - 1]], [gpb[num_levels - 1]], [gpm[num_levels - 1]]
    for i in range(num_levels - 1, 0, -1):
        lpa.append(gpa[i - 1].astype(np.float32) - pyr_up(gpa[i]))
        lpb.append(gpb[i - 1].astype(np.float32) - pyr_up(gpb[i]))
        gmr.append(gpm[i - 1])
    ls = [la * g + lb * (1.0 - g) for la, lb, g in zip(lpa, lpb, gmr)]
    out = ls[0]
    for i in range(1, num_levels):
        out = pyr_up(out) + ls[i]
    return out


def blending(full_img_u8: np.ndarray, ori_img: np.ndarray, mask: np.ndarray) -> np.ndarray:
    """``blending`` (multi_band_blending.py:51-74) for 1024 x 1024 inputs (its two cv2.resize calls are then identities): uint8 [H,W,3]."""
    assert full_img_u8.shape[:2] == (1024, 1024) and ori_img.shape[:2] == (1024, 1024), "restated for the 1024 x 1024 call site"
    img = laplacian_blend(full_img_u8, ori_img.astype(np.float64), mask.astype(np.float32), 10)
    return np.clip(img, 0, 255).astype(np.uint8)


# ------------------------------------------------------------------------------------ f3: PIL's default resize of the swapped face
# face_swap_video_pipeline.py:447 — ``swapped_face_image.resize((512, 512)).resize((1024, 1024))`` — is Pillow's ``Image.resize`` with
# its default filter for RGB images, BICUBIC (Pillow==10.1.0 in requirements.txt:134; src/libImaging/Resample.c, unchanged in the
# Pillow 12 of this image).  Restated in integers exactly as the library computes it; PINNED: tests/test_oracle_golden.py compares it
# with PIL itself on ragged sizes, up- and down-scaling.
_PIL_PRECISION_BITS = 32 - 8 - 2


def _pil_bicubic(x: np.ndarray) -> np.ndarray:
    a = -0.5
    x = np.abs(x)
    return np.where(x < 1.0, ((a + 2.0) * x - (a + 3.0)) * x * x + 1.0, np.where(x < 2.0, (((x - 5.0) * x + 8.0) * x - 4.0) * a, 0.0))


def pil_resample_coeffs(in_size: int, out_size: int):
    """``precompute_coeffs`` + ``normalize_coeffs_8bpc`` (Resample.c) for the bicubic filter over the whole axis:
    ``(xmin [out], count [out], k int32 [out, ksize])``."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    xmin = np.zeros(out_size, np.int32)
    cnt = np.zeros(out_size, np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        lo = max(int(center - support + 0.5), 0)
        hi = min(int(center + support + 0.5), in_size)
        n = hi - lo
        w = _pil_bicubic((np.arange(n) + lo - center + 0.5) * ss)
        tot = w.sum()
        if tot != 0.0:
            w = w / tot
        scaled = w * float(1 << _PIL_PRECISION_BITS)
        kk[xx, :n] = np.where(w < 0, np.trunc(-0.5 + scaled), np.trunc(0.5 + scaled)).astype(np.int64)
        xmin[xx], cnt[xx] = lo, n
    return xmin, cnt, kk


def _pil_resample_axis(img: np.ndarray, out_size: int, axis: int) -> np.ndarray:
    xmin, cnt, kk = pil_resample_coeffs(img.shape[axis], out_size)
    a = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((out_size,) + a.shape[1:], np.uint8)
    for xx in range(out_size):
        acc = np.tensordot(kk[xx, :cnt[xx]].astype(np.int64), a[xmin[xx]:xmin[xx] + cnt[xx]], axes=(0, 0)) + (1 << (_PIL_PRECISION_BITS - 1))
        out[xx] = np.clip(acc >> _PIL_PRECISION_BITS, 0, 255)
    return np.moveaxis(out, 0, axis)


def pil_resize_bicubic(img_u8: np.ndarray, size) -> np.ndarray:
    """``PIL.Image.fromarray(img).resize(size)`` (size = (width, height), default BICUBIC) of a uint8 ``[H, W, C]`` image: a horizontal then
    a vertical pass, each rounded to 8 bits (ImagingResample)."""
    wd, ht = size
    out = img_u8
    if wd != img_u8.shape[1]:
        out = _pil_resample_axis(out, wd, 1)
    if ht != img_u8.shape[0]:
        out = _pil_resample_axis(out, ht, 0)
    return out
