"""Drop-in for the reference's ``models/stylegan2/model.py`` (synthesis side only).

Same class names, constructor arguments, ``forward`` signatures, parameter/buffer names and shapes as the reference
(``Generator`` :482-698, ``StyledConv`` :351-423, ``ToRGB`` :426-479, ``ModulatedConv2d`` :184-320, ``NoiseInjection``
:323-335, ``ConstantInput`` :338-348, ``Blur`` :78-94, ``Upsample`` :34-53, ``EqualLinear`` :135-164, ``PixelNorm`` :15-20),
so reference checkpoints load unchanged.  What differs is *how* a masked layer is evaluated: the reference loops over the
12 regions, running a full modulated convolution per region and summing ``out_i * segmap_i`` (:395-398, :451-454); here
one region-aware HIP kernel per layer looks up the region of every output pixel and applies that region's modulation and
demodulation inside a single implicit-GEMM pass (``e4s2024_amd/csrc/modconv.hip``).

Out of scope (training only): ``Discriminator``, ``ConvLayer``, ``ResBlock``, ``Downsample``, ``EqualConv2d``.
"""
import math
import random

import torch
from torch import nn

from e4s2024_amd import ops, torch_ref
from models.stylegan2.op import FusedLeakyReLU, fused_leaky_relu, upfirdn2d


class PixelNorm(nn.Module):
    """reference :15-20.  Only reached with ``input_is_latent=False`` (mapping network), which no caller of the RGI path uses."""

    def __init__(self):
        super().__init__()

    def forward(self, input):
        return input * torch.rsqrt(torch.mean(input ** 2, dim=1, keepdim=True) + 1e-8)


def make_kernel(k):
    k = torch.tensor(k, dtype=torch.float32)
    if k.ndim == 1:
        k = k[None, :] * k[:, None]
    k /= k.sum()
    return k


class Upsample(nn.Module):
    """reference :34-53 — FIR x2 upsample of the RGB skip.  Inside ``ToRGB`` it is fused into the ToRGB kernel; called on
    its own it runs the stand-alone upfirdn2d kernel."""

    def __init__(self, kernel, factor=2):
        super().__init__()
        self.factor = factor
        kernel = make_kernel(kernel) * (factor ** 2)
        self.register_buffer("kernel", kernel)
        p = kernel.shape[0] - factor
        pad0 = (p + 1) // 2 + factor - 1
        pad1 = p // 2
        self.pad = (pad0, pad1)

    def forward(self, input):
        return upfirdn2d(input, self.kernel, up=self.factor, down=1, pad=self.pad)


class Blur(nn.Module):
    """reference :78-94.  Inside an up-sampling ``ModulatedConv2d`` the blur is composed into the convolution weights."""

    def __init__(self, kernel, pad, upsample_factor=1):
        super().__init__()
        kernel = make_kernel(kernel)
        if upsample_factor > 1:
            kernel = kernel * (upsample_factor ** 2)
        self.register_buffer("kernel", kernel)
        self.pad = pad

    def forward(self, input):
        return upfirdn2d(input, self.kernel, pad=self.pad)


class EqualLinear(nn.Module):
    """reference :135-164."""

    def __init__(self, in_dim, out_dim, bias=True, bias_init=0, lr_mul=1, activation=None):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(out_dim, in_dim).div_(lr_mul))
        if bias:
            self.bias = nn.Parameter(torch.zeros(out_dim).fill_(bias_init))
        else:
            self.bias = None
        self.activation = activation
        self.scale = (1 / math.sqrt(in_dim)) * lr_mul
        self.lr_mul = lr_mul

    def forward(self, input):
        shp = input.shape
        x = input.reshape(-1, 1, shp[-1])
        out = ops.grouped_linear(x, [self.weight], None if self.bias is None else [self.bias], scale=self.scale, bias_mul=self.lr_mul,
                                 act=2 if self.activation else 0, slope=0.2)
        out = out.reshape(*shp[:-1], self.weight.shape[0])
        return ops._attach("EqualLinear", out, input, self.weight, self.bias,
                           ref=lambda x, w, b: torch_ref.equal_linear(x, w, b, self.scale, self.lr_mul, bool(self.activation)))

    def __repr__(self):
        return f"{self.__class__.__name__}({self.weight.shape[1]}, {self.weight.shape[0]})"


class ModulatedConv2d(nn.Module):
    """reference :184-320 (fused branch).  ``forward(input, style)`` with ``style [bs, 512]`` is the plain (single-region)
    call; ``forward_regions`` is the one-pass masked form used by ``StyledConv`` / ``ToRGB``."""

    def __init__(self, in_channel, out_channel, kernel_size, style_dim, demodulate=True, upsample=False, downsample=False,
                 blur_kernel=[1, 3, 3, 1], fused=True):
        super().__init__()
        if downsample:
            raise NotImplementedError("downsample=True is only used by the training-only Discriminator (out of scope)")
        if kernel_size not in (1, 3):
            raise NotImplementedError("kernel_size must be 1 or 3")
        if kernel_size == 1 and (upsample or out_channel != 3 or demodulate):
            raise NotImplementedError("1x1 modulated conv is supported in its ToRGB form (3 outputs, no demodulation)")
        self.eps = 1e-8
        self.kernel_size = kernel_size
        self.in_channel = in_channel
        self.out_channel = out_channel
        self.upsample = upsample
        self.downsample = downsample
        if upsample:
            factor = 2
            p = (len(blur_kernel) - factor) - (kernel_size - 1)
            pad0 = (p + 1) // 2 + factor - 1
            pad1 = p // 2 + 1
            if len(blur_kernel) != 4:
                raise NotImplementedError("up-conv blur must have 4 taps")
            self.blur = Blur(blur_kernel, pad=(pad0, pad1), upsample_factor=factor)
        fan_in = in_channel * kernel_size ** 2
        self.scale = 1 / math.sqrt(fan_in)
        self.padding = kernel_size // 2
        self.weight = nn.Parameter(torch.randn(1, out_channel, in_channel, kernel_size, kernel_size))
        self.modulation = EqualLinear(style_dim, in_channel, bias_init=1)
        self.demodulate = demodulate
        self.fused = fused
        self._prepared = ops.PreparedWeights()
        self._prepared_tconv = ops.PreparedWeights()
        self._prepared_mx = ops.PreparedMx()
        self._prepared_mx4 = ops.PreparedMx()      # the same weight as the four-parity up kernel reads it (arith 4)
        self._prepared_ubmx = ops.PreparedMx()     # ... as the region-uniform block kernel reads it (arith 7: csrc/modconv_upblock_mx.hip)
        self._prepared_hc = ops.PreparedHc()

    def __repr__(self):
        return (f"{self.__class__.__name__}({self.in_channel}, {self.out_channel}, {self.kernel_size}, "
                f"upsample={self.upsample}, downsample={self.downsample})")

    def _two_stage(self, masked):
        return self.upsample and not masked and ops.MODCONV_MODE == "sb" and ops.UP_TWO_STAGE

    def _weights(self, masked):
        """Prepared weight slabs (+ the squared-sum table for demodulation) for the kernel this layer will run."""
        if self._two_stage(masked):
            return self._prepared_tconv.get(self.weight, None, False, self.demodulate, tconv=True)
        return self._prepared.get(self.weight, self.blur.kernel if self.upsample else None, self.upsample, self.demodulate)

    def table_job(self, styles, masked):
        """Entry for ``ops.style_demod_plan``: this layer's (s, d) tables are then computed together with every other layer's."""
        _, wsq = self._weights(masked)
        return (id(self), styles, self.modulation.weight, self.modulation.bias, wsq, self.out_channel)

    def _tables(self, styles, wsq):
        planned = ops.style_demod_planned(id(self), styles)
        if planned is not None:
            return planned
        return ops.style_demod(styles, self.modulation.weight, self.modulation.bias, wsq, self.out_channel)

    def tables(self, styles, masked=True):
        """styles ``[bs, nreg, 512]`` → (wt, s, d) device tables for this layer."""
        wt, wsq = self._weights(masked)
        s, d = self._tables(styles, wsq)
        return wt, s, d

    def accepts_nhwc(self, masked):
        """Can this layer read a channels-last activation?  (the single-region layers on the split-bf16 kernels; engine-internal)"""
        if masked or self.kernel_size != 3 or ops.MODCONV_MODE != "sb" or self.in_channel % 16 or self.out_channel % 8:
            return False
        return (ops.UP_FUSED and self._two_stage(False)) if self.upsample else True

    def forward_regions(self, input, styles, labels, noise=None, noise_weight=None, act_bias=None, act=False, rgb=None, want_out=True,
                        x_nhwc=False, out_nhwc=False, x_sp=False, s_next=None):
        """``x_sp`` / ``s_next`` (engine-internal, inference): the split-plane chain of the single-region stages (csrc/modconv_chain.hip) —
        ``input`` is split planes already carrying this layer's modulation / the activation is written as split planes modulated by
        ``s_next`` for the next layer."""
        if x_sp:
            if labels is not None or self.kernel_size != 3:
                raise ValueError("split-plane input is built for the single-region 3x3 layers")
            wt, s, d = self.tables(styles, masked=False)
            if self.upsample:
                hc = self._prepared_hc.get(self.weight, self.blur.kernel) if d is not None else None      # half-composed form (csrc/modconv_uphc.hip)
                return ops.modconv_up_single(input, wt, s, d, self.blur.kernel, noise, noise_weight, act_bias, act, self.out_channel, s_next=s_next, hc=hc)
            out_sp, rgb_img = ops.chain_conv3x3(input, wt, d, noise, noise_weight, act_bias, act, self.out_channel, s_next=s_next, rgb=rgb)
            return out_sp if rgb is None else (out_sp, rgb_img)
        if self._two_stage(labels is not None):
            # single-region up layer: transposed conv at 1x its MACs into a pre-blur buffer, then blur + epilogue
            wt, s, d = self.tables(styles, masked=False)
            out = ops.modconv_up_single(input, wt, s, d, self.blur.kernel, noise, noise_weight, act_bias, act, self.out_channel,
                                        x_nhwc=x_nhwc, out_nhwc=out_nhwc)
            if x_nhwc or out_nhwc:
                return out          # inference only (Generator.forward takes this route under no_grad)
            return ops._attach("ModulatedConv2d", out, input, styles, self.weight, self.modulation.weight, self.modulation.bias, noise_weight,
                               act_bias, ref=self._torch_ref(labels, noise, act))
        wt, s, d = self.tables(styles, masked=labels is not None)
        saved = (s, d, self._weights(labels is not None)[1]) if labels is not None else None
        up_blocks = None
        if (self.upsample and labels is not None and ops.UP_BLOCKS and ops.MODCONV_MODE == "sb" and self.kernel_size == 3 and ops.mx_arith() == 1
                and input.shape[-1] >= max(32, ops.UP_BLOCKS_MIN_WIDTH) and self.out_channel >= 128 and self.in_channel % 32 == 0 and self.in_channel <= 512
                and not torch.is_grad_enabled()):
            # region-uniform 16 x 16 output blocks run in the transposed-conv form on f16 + fp6 (csrc/modconv_upblock_mx.hip: a third preparation of the same weight)
            up_blocks = (self._prepared_ubmx.get(self.weight, None, False, 7), self.blur.kernel)
        mx = mx4 = None
        if (self.kernel_size == 3 and not (x_nhwc or out_nhwc) and isinstance(wt, tuple)
                and ops.mx_eligible(self.in_channel, self.out_channel, input.shape[-1], labels is not None)):
            arith = ops.mx_arith()         # the DMA-fed masked kernel (csrc/modconv_mx.hip), inference only
            mx = (self._prepared_mx.get(self.weight, self.blur.kernel if self.upsample else None, self.upsample, arith), arith)
            if (self.upsample and arith == 1 and up_blocks is None and rgb is None
                    and ops.mx4_eligible(self.in_channel, self.out_channel, input.shape[-2], input.shape[-1], input.shape[0])):
                mx4 = self._prepared_mx4.get(self.weight, self.blur.kernel, True, 4)      # the four-parity kernel's copy (csrc/modconv_mx4.hip)
        out = ops.region_modconv3x3(input, wt, s, d, labels, noise, noise_weight, act_bias, act, self.out_channel, self.upsample, rgb=rgb,
                                    want_out=want_out, x_nhwc=x_nhwc, out_nhwc=out_nhwc, s_next=s_next, up_blocks=up_blocks, mx=mx, mx4=mx4)
        if x_nhwc or out_nhwc or s_next is not None:
            return out              # inference only (Generator.forward takes this route under no_grad)
        if rgb is not None:
            return tuple(None if o is None else ops._attach("ModulatedConv2d", o, input, styles, self.weight, self.modulation.weight,
                                                            self.modulation.bias, noise_weight, act_bias) for o in out)
        return ops._attach("ModulatedConv2d", out, input, styles, self.weight, self.modulation.weight, self.modulation.bias, noise_weight,
                           act_bias, ref=self._torch_ref(labels, noise, act, saved))

    def _torch_ref(self, labels, noise, act, tables=None):
        """The differentiable PyTorch form of ``forward_regions`` for the backward pass (torch_ref.styled_conv)."""
        blur = self.blur.kernel if self.upsample else None
        mod = self.modulation

        def ref(x, styles, weight, mod_w, mod_b, noise_weight, act_bias, fwd_out=None):
            return torch_ref.styled_conv(x, styles, weight, mod_w, mod_b, noise_weight, act_bias, labels=labels, noise=noise, act=act,
                                         upsample=self.upsample, blur=blur, demodulate=self.demodulate, mod_scale=mod.scale, mod_lr=mod.lr_mul,
                                         fwd_out=fwd_out, tables=tables)
        ref.takes_fwd_out = True      # the backward differentiates from the kernel's own output: the layer is not re-evaluated
        return ref

    def forward(self, input, style):
        if self.kernel_size == 1:
            wt, s, _ = self.tables(style[:, None, :])
            out = ops.region_torgb(input, wt, s, None, torch.zeros(3, device=input.device), None, None)
            mod = self.modulation
            return ops._attach("ModulatedConv2d", out, input, style, self.weight, self.modulation.weight, self.modulation.bias,
                               ref=lambda x, st, w, mw, mb: torch_ref.to_rgb(x, st[:, None, :], None, w, mw, mb, 0.0, labels=None, up_kernel=None,
                                                                              mod_scale=mod.scale, mod_lr=mod.lr_mul))
        return self.forward_regions(input, style[:, None, :], None)


class NoiseInjection(nn.Module):
    """reference :323-335.  Inside ``StyledConv`` the injection is fused into the conv epilogue."""

    def __init__(self):
        super().__init__()
        self.weight = nn.Parameter(torch.zeros(1))

    def forward(self, image, noise=None):
        if noise is None:
            batch, _, height, width = image.shape
            noise = image.new_empty(batch, 1, height, width).normal_()
        return image + self.weight * noise


class ConstantInput(nn.Module):
    """reference :338-348."""

    def __init__(self, channel, size=4):
        super().__init__()
        self.input = nn.Parameter(torch.randn(1, channel, size, size))

    def forward(self, input):
        batch = input.shape[0]
        return self.input.repeat(batch, 1, 1, 1)


class StyledConv(nn.Module):
    """reference :351-423."""

    def __init__(self, in_channel, out_channel, kernel_size, style_dim, upsample=False, blur_kernel=[1, 3, 3, 1], demodulate=True,
                 mask_op=False):
        super().__init__()
        self.conv = ModulatedConv2d(in_channel, out_channel, kernel_size, style_dim, upsample=upsample, blur_kernel=blur_kernel,
                                    demodulate=demodulate)
        self.noise = NoiseInjection()
        self.activate = FusedLeakyReLU(out_channel)
        self.mask_op = mask_op

    def forward(self, input, style, mask, noise=None, _fused_rgb=None, _want_out=True, _x_nhwc=False, _out_nhwc=False, _x_sp=False, _s_next=None):
        """``_fused_rgb=(to_rgb, rgb_style [bs,512], skip)`` (engine-internal, used by ``Generator.forward``) also evaluates that
        single-region ToRGB in this layer's epilogue and returns ``(out, rgb)``.  ``_x_nhwc`` / ``_out_nhwc`` (engine-internal): the
        activation comes in / goes out channels-last, ``[bs, H, W, C]``."""
        if _x_sp:
            _, bs, _, H, W, _ = input.shape       # split planes [2, bs, C/8, H, W, 8]
        elif _x_nhwc:
            bs, _, H, W, _ = input.shape          # channel-blocked [bs, C/8, H, W, 8]
        else:
            bs, _, H, W = input.shape
        H_out, W_out = (H * 2, W * 2) if self.conv.upsample else (H, W)
        if noise is None:  # reference :331-333
            noise = torch.empty(bs, 1, H_out, W_out, dtype=torch.float32, device=input.device).normal_()
        if self.activate.negative_slope != 0.2 or abs(self.activate.scale - 2 ** 0.5) > 1e-12:
            raise NotImplementedError("fused epilogue implements leaky_relu(0.2) * sqrt(2)")
        if self.mask_op:
            labels = ops.mask_to_labels(mask)          # [bs, h, w] uint8, cached per mask object
            styles = style                              # [bs, n_regions, 512]
        else:
            if style.dim() != 2:
                raise ValueError(f"unmasked StyledConv expects style [bs, 512], got {tuple(style.shape)}")
            labels = None
            styles = style[:, None, :]
        rgb = None
        if _fused_rgb is not None:
            to_rgb, rgb_style, skip = _fused_rgb
            r_wt, r_s, _ = to_rgb.conv.tables(rgb_style[:, None, :])
            rgb = (r_wt, r_s, to_rgb.bias, skip, to_rgb.upsample.kernel if skip is not None else None)
        return self.conv.forward_regions(input, styles, labels, noise, self.noise.weight, self.activate.bias, act=True, rgb=rgb,
                                         want_out=_want_out or rgb is None, x_nhwc=_x_nhwc, out_nhwc=_out_nhwc, x_sp=_x_sp, s_next=_s_next)


class ToRGB(nn.Module):
    """reference :426-479."""

    def __init__(self, in_channel, style_dim, upsample=True, blur_kernel=[1, 3, 3, 1], mask_op=False):
        super().__init__()
        if upsample:
            self.upsample = Upsample(blur_kernel)
        self.conv = ModulatedConv2d(in_channel, 3, 1, style_dim, demodulate=False)
        self.bias = nn.Parameter(torch.zeros(1, 3, 1, 1))
        self.mask_op = mask_op

    def forward(self, input, style, mask, skip=None):
        if self.mask_op:
            labels = ops.mask_to_labels(mask)
            styles = style
        else:
            if style.dim() != 2:
                raise ValueError(f"unmasked ToRGB expects style [bs, 512], got {tuple(style.shape)}")
            labels = None
            styles = style[:, None, :]
        wt, s, _ = self.conv.tables(styles)
        fuse_skip = skip is not None and tuple(self.upsample.kernel.shape) == (4, 4) and self.upsample.factor == 2
        out = ops.region_torgb(input, wt, s, labels, self.bias, skip if fuse_skip else None, self.upsample.kernel if fuse_skip else None)
        if skip is not None and not fuse_skip:
            out = out + self.upsample(skip)
        mod = self.conv.modulation
        up_kernel = self.upsample.kernel if skip is not None else None
        if skip is not None and (tuple(up_kernel.shape) != (4, 4) or self.upsample.factor != 2):
            raise NotImplementedError("ToRGB backward is written for the 4x4, factor-2 skip upsample")

        def ref(x, st, sk, w, mw, mb, bias, fwd_out=None):
            return torch_ref.to_rgb(x, st if self.mask_op else st[:, None, :], sk, w, mw, mb, bias, labels=labels, up_kernel=up_kernel,
                                    mod_scale=mod.scale, mod_lr=mod.lr_mul, fwd_out=fwd_out, tables=(s, None, None))
        ref.takes_fwd_out = True
        return ops._attach("ToRGB", out, input, style, skip, self.conv.weight, self.conv.modulation.weight, self.conv.modulation.bias, self.bias,
                           ref=ref)


class Generator(nn.Module):
    """reference :482-698."""

    def __init__(self, size, style_dim, n_mlp, channel_multiplier=2, blur_kernel=[1, 3, 3, 1], lr_mlp=0.01, split_layer_idx=7,
                 remaining_layer_idx=18):
        super().__init__()
        self.split_layer_idx = split_layer_idx
        self.remaining_layer_idx = remaining_layer_idx
        self.size = size
        self.style_dim = style_dim

        layers = [PixelNorm()]
        for i in range(n_mlp):
            layers.append(EqualLinear(style_dim, style_dim, lr_mul=lr_mlp, activation="fused_lrelu"))
        self.style = nn.Sequential(*layers)

        self.channels = {
            4: 512, 8: 512, 16: 512, 32: 512,
            64: 256 * channel_multiplier, 128: 128 * channel_multiplier, 256: 64 * channel_multiplier,
            512: 32 * channel_multiplier, 1024: 16 * channel_multiplier,
        }
        self.input = ConstantInput(self.channels[4])
        self.conv1 = StyledConv(self.channels[4], self.channels[4], 3, style_dim, blur_kernel=blur_kernel, mask_op=True)
        self.to_rgb1 = ToRGB(self.channels[4], style_dim, upsample=False, mask_op=True)

        self.log_size = int(math.log(size, 2))
        self.num_layers = (self.log_size - 2) * 2 + 1

        self.convs = nn.ModuleList()
        self.upsamples = nn.ModuleList()
        self.to_rgbs = nn.ModuleList()
        self.noises = nn.Module()

        in_channel = self.channels[4]
        for layer_idx in range(self.num_layers):
            res = (layer_idx + 5) // 2
            self.noises.register_buffer(f"noise_{layer_idx}", torch.randn(1, 1, 2 ** res, 2 ** res))

        for i in range(3, self.log_size + 1):
            out_channel = self.channels[2 ** i]
            masked = not (i > (2 + self.remaining_layer_idx // 2))
            self.convs.append(StyledConv(in_channel, out_channel, 3, style_dim, upsample=True, blur_kernel=blur_kernel, mask_op=masked))
            self.convs.append(StyledConv(out_channel, out_channel, 3, style_dim, blur_kernel=blur_kernel, mask_op=masked))
            self.to_rgbs.append(ToRGB(out_channel, style_dim,
                                      mask_op=not (self.remaining_layer_idx != 17 and i >= (2 + self.remaining_layer_idx // 2))))
            in_channel = out_channel

        self.n_latent = self.log_size * 2 - 2

    def _table_jobs(self, latent, lat=None):
        """(layer, W+ slice) pairs in the order ``forward`` visits them — reference :661-690.  ``lat``: the per-index code tensors ``forward`` hands to the layers (under
        autograd they are slices of a transposed COPY of ``latent``): the plan is keyed on the very tensor a layer will ask with, so the jobs must be built from those —
        built from ``latent`` itself the plan missed under autograd and every layer launched its own two table kernels (43 launches of a PTI step, round 6)."""
        rli = self.remaining_layer_idx

        def job(layer, masked, idx):
            if lat is not None:
                st = lat[idx] if masked else lat[idx][:, 0][:, None, :]
            else:
                st = latent[:, :, idx] if masked else latent[:, 0, idx][:, None, :]
            return layer.conv.table_job(st, masked)
        jobs = [job(self.conv1, True, 0), job(self.to_rgb1, True, 1)]
        for j, to_rgb in enumerate(self.to_rgbs):
            i = 2 * j + 1
            per_region = i < rli
            for conv, idx in ((self.convs[2 * j], i), (self.convs[2 * j + 1], i + 1)):
                if conv.mask_op != per_region:
                    return []          # inconsistent configuration (even remaining_layer_idx): let the per-layer path raise
                jobs.append(job(conv, per_region, idx))
            single = (not per_region) or (rli != 17 and i + 2 == rli)
            if to_rgb.mask_op == single:
                return []
            jobs.append(job(to_rgb, not single, i + 2))
        return jobs

    def make_noise(self):
        device = self.input.input.device
        noises = [torch.randn(1, 1, 2 ** 2, 2 ** 2, device=device)]
        for i in range(3, self.log_size + 1):
            for _ in range(2):
                noises.append(torch.randn(1, 1, 2 ** i, 2 ** i, device=device))
        return noises

    def mean_latent(self, n_latent):
        latent_in = torch.randn(n_latent, self.style_dim, device=self.input.input.device)
        return self.style(latent_in).mean(0, keepdim=True)

    def get_latent(self, input):
        return self.style(input)

    def forward(self, styles, structure_feats, mask, return_latents=False, inject_index=None, truncation=1, truncation_latent=None,
                input_is_latent=False, noise=None, randomize_noise=True, use_structure_code=False):
        # ops.guarded: the masked 3x3 layers run in f16 (+ fp6) arithmetic at inference; a pass that left the f16 range is re-run once in split-bf16
        # (one host synchronisation per call, which the reference's callers do right behind gen_img anyway; pipelines own the guard themselves)
        with ops.one_forward():        # parameters do not change inside one pass: a trained layer's weights are re-laid out once per pass
            return ops.guarded(lambda: self._forward(styles, structure_feats, mask, return_latents, inject_index, truncation, truncation_latent,
                                                     input_is_latent, noise, randomize_noise, use_structure_code))

    def _forward(self, styles, structure_feats, mask, return_latents=False, inject_index=None, truncation=1, truncation_latent=None,
                 input_is_latent=False, noise=None, randomize_noise=True, use_structure_code=False):
        if not input_is_latent:
            styles = [self.style(s) for s in styles]

        if noise is None:
            if randomize_noise:
                noise = [None] * self.num_layers
                st0 = styles[0] if isinstance(styles, (list, tuple)) else styles
                if torch.is_tensor(st0) and st0.is_cuda:
                    # one normal_() for every layer's fresh noise map (reference :331-333 draws one per layer: 17 launches per pass — in the PTI step they are 17 of
                    # its ~600 launches); layer i's map is a view [bs, 1, H, W] of the flat draw
                    bs_n = st0.shape[0]
                    shapes = [tuple(getattr(self.noises, f"noise_{i}").shape[-2:]) for i in range(self.num_layers)]
                    flat = torch.empty(bs_n * sum(h * w for h, w in shapes), dtype=torch.float32, device=st0.device).normal_()
                    o = 0
                    for i, (h, w) in enumerate(shapes):
                        noise[i] = flat[o:o + bs_n * h * w].view(bs_n, 1, h, w)
                        o += bs_n * h * w
            else:
                noise = [getattr(self.noises, f"noise_{i}") for i in range(self.num_layers)]

        if truncation < 1:
            styles = [truncation_latent + truncation * (style - truncation_latent) for style in styles]

        if len(styles) < 2:
            inject_index = self.n_latent
            if styles[0].ndim < 4:
                latent = styles[0].unsqueeze(1).repeat(1, inject_index, 1)
            else:
                latent = styles[0]
        else:
            if inject_index is None:
                inject_index = random.randint(1, self.n_latent - 1)
            latent = styles[0].unsqueeze(1).repeat(1, inject_index, 1)
            latent2 = styles[1].unsqueeze(1).repeat(1, self.n_latent - inject_index, 1)
            latent = torch.cat([latent, latent2], 1)

        rli = self.remaining_layer_idx
        # one view per W+ index: under autograd the 18 views are one unbind (its backward one stack), not 26 zero-filled slice gradients
        if latent.ndim == 4 and latent.is_cuda and torch.is_grad_enabled():
            # (the layers' gradient kernels want dense [bs, regions, 512] codes: one transposed copy here instead of one per layer)
            lat = latent.permute(2, 0, 1, 3).contiguous().unbind(0)
        else:
            lat = latent.unbind(2) if latent.ndim == 4 else None
        if latent.ndim == 4 and latent.is_cuda:
            ops.style_demod_plan(self._table_jobs(latent, lat))     # every layer's modulation / demodulation table in two launches
        out = self.input(latent)
        out = self.conv1(out, lat[0] if lat is not None else latent[:, :, 0], mask, noise=noise[0])
        skip = self.to_rgb1(out, lat[1] if lat is not None else latent[:, :, 1], mask)

        intermediate_feats = None
        # Channels-last chain (inference): from the last layer before the single-region stages on, activations stay [bs, H, W, C] between the
        # fused kernels — a tile's halo is then shared by all channels of a pixel instead of costing three cache lines per channel row
        # (measured: 1.8 -> 5 TB/s on the staging read pattern, tools/probes/tile_read_probe.hip).  Nothing outside these kernels sees it.
        chain = ops.NHWC_CHAIN and not torch.is_grad_enabled() and out.is_cuda and latent.ndim == 4 and ops.FUSE_RGB
        nhwc = False                           # layout of `out` right now
        # Split-plane chain (inference): from stage sp_from on, every layer is a single-region layer with a split-plane kernel; the activation
        # between them is written by its producer already multiplied by the consumer's modulation and split into bf16 hi / lo planes
        # (csrc/modconv_chain.hip).  The last masked layer hands over in that form too.
        n_stage = len(self.to_rgbs)

        def sp_stage_ok(jj):
            cu, c2, tr = self.convs[2 * jj], self.convs[2 * jj + 1], self.to_rgbs[jj]
            res = 2 ** (jj + 3)
            return (2 * jj + 1 >= rli and not cu.mask_op and not c2.mask_op and not tr.mask_op and tuple(tr.upsample.kernel.shape) == (4, 4)
                    and cu.conv.kernel_size == 3 and c2.conv.kernel_size == 3 and cu.conv._two_stage(False)
                    and ops.chain_supported(cu.conv.in_channel, cu.conv.out_channel, res // 2, res // 2, True)
                    and ops.chain_supported(c2.conv.in_channel, c2.conv.out_channel, res, res, False, last=(jj == n_stage - 1))
                    and ops.can_fuse_rgb(c2.conv.out_channel, res, False, False))
        sp_from = n_stage
        if chain and ops.SP_CHAIN and lat is not None and not torch.is_grad_enabled():
            while sp_from > 0 and sp_stage_ok(sp_from - 1):
                sp_from -= 1
        sp = False                             # `out` is split planes right now

        def s_of(layer, k):                    # modulation table [bs, 1, cin] a single-region layer will apply to its input (W+ index k)
            return layer.conv.tables(lat[k][:, 0][:, None, :], masked=False)[1]

        def up_takes_nhwc(jj):                 # may the up-conv of stage jj read channels-last?
            if not chain or jj >= len(self.to_rgbs) or 2 * jj + 1 < rli or not ops.nhwc_link("u", jj):
                return False
            cu = self.convs[2 * jj]
            return (not cu.mask_op) and cu.conv.accepts_nhwc(False)

        for j, to_rgb in enumerate(self.to_rgbs):
            i = 2 * j + 1                      # W+ index shared by to_rgbs[j-1] and this resolution's up-conv
            per_region = i < rli               # reference :670 — below it every layer receives one code per region
            if lat is not None:
                code = (lambda k: lat[k]) if per_region else (lambda k: lat[k][:, 0])
            else:
                code = (lambda k: latent[:, :, k]) if per_region else (lambda k: latent[:, 0, k])
            conv2 = self.convs[2 * j + 1]
            if j >= sp_from:
                # ---- a stage of the split-plane chain: up-conv -> conv (+ fused ToRGB), hand-overs as split planes
                if not sp:                     # (the previous layer could not hand over in split planes: convert once)
                    out = ops.to_split_planes(out, s_of(self.convs[2 * j], i), x_nhwc=nhwc)
                    nhwc, sp = False, True
                out = self.convs[2 * j](out, code(i), mask, noise=noise[1 + 2 * j], _x_sp=True, _s_next=s_of(conv2, i + 1))
                last = j + 1 == n_stage
                out, skip = conv2(out, code(i + 1), mask, noise=noise[2 + 2 * j], _fused_rgb=(to_rgb, lat[i + 2][:, 0], skip), _want_out=not last,
                                  _x_sp=True, _s_next=None if last else s_of(self.convs[2 * j + 2], i + 2))
                continue
            # this stage's second conv can take channels-last input iff it is a single-region layer whose ToRGB rides in its epilogue
            c2_fused = ((not per_region or (rli != 17 and i + 2 == rli)) and not to_rgb.mask_op and tuple(to_rgb.upsample.kernel.shape) == (4, 4)
                        and ops.can_fuse_rgb(conv2.conv.out_channel, out.shape[3 if nhwc else -1] * 2, False, conv2.mask_op))
            c2_nhwc_in = chain and c2_fused and not conv2.mask_op and conv2.conv.accepts_nhwc(False) and ops.nhwc_link("c", j)
            if nhwc or c2_nhwc_in:
                out = self.convs[2 * j](out, code(i), mask, noise=noise[1 + 2 * j], _x_nhwc=nhwc, _out_nhwc=c2_nhwc_in)
                nhwc = c2_nhwc_in
            else:
                out = self.convs[2 * j](out, code(i), mask, noise=noise[1 + 2 * j])
            if per_region and i + 2 == self.split_layer_idx:   # reference :673-678
                if use_structure_code:
                    out = structure_feats
                intermediate_feats = out
            single = (not per_region) or (rli != 17 and i + 2 == rli)   # reference :681-688
            # (inference only: the fused pair has no backward form — under autograd the two layers run separately)
            needs_grad = torch.is_grad_enabled() and (out.requires_grad or latent.requires_grad or any(p.requires_grad for p in conv2.parameters())
                                                      or any(p.requires_grad for p in to_rgb.parameters()))
            if (single and not needs_grad and not to_rgb.mask_op and out.is_cuda and tuple(to_rgb.upsample.kernel.shape) == (4, 4)
                    and ops.can_fuse_rgb(conv2.conv.out_channel, out.shape[3 if nhwc else -1], False, conv2.mask_op)):
                # the single-region ToRGB rides in the conv's epilogue: the activation is not read back for the 1x1 conv
                # (the last layer's own activation is consumed by nothing but this ToRGB: it is not written)
                last = j + 1 == len(self.to_rgbs)
                # the last layer before the split-plane chain hands over in that form (masked fused-ToRGB kernel; otherwise converted there)
                sp_out = (j + 1 == sp_from and not last and not nhwc and conv2.mask_op and conv2.conv.out_channel % 8 == 0 and ops.MODCONV_MODE == "sb"
                          and out.shape[-1] >= 32)
                nhwc_out = (not last) and not sp_out and up_takes_nhwc(j + 1) and conv2.conv.out_channel % 8 == 0
                out, skip = conv2(out, code(i + 1), mask, noise=noise[2 + 2 * j], _fused_rgb=(to_rgb, lat[i + 2][:, 0] if lat is not None else latent[:, 0, i + 2], skip),
                                  _want_out=not last, _x_nhwc=nhwc, _out_nhwc=nhwc_out,
                                  _s_next=s_of(self.convs[2 * j + 2], i + 2) if sp_out else None)
                nhwc, sp = nhwc_out, sp_out
            else:
                assert not nhwc, "channels-last activation reached a layer that cannot read it"
                out = conv2(out, code(i + 1), mask, noise=noise[2 + 2 * j])
                if lat is not None:
                    skip = to_rgb(out, lat[i + 2][:, 0] if single else lat[i + 2], mask, skip)
                else:
                    skip = to_rgb(out, latent[:, 0, i + 2] if single else latent[:, :, i + 2], mask, skip)

        ops.table_plan_clear()
        image = skip
        if return_latents:
            return image, latent, intermediate_feats
        return image, None, intermediate_feats
